#!/usr/bin/env python3
"""fastq_count on multi-member gzip: wall time vs inflate threads per file (HPN_GZ_THREADS), 1 and 8 files."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, REF = os.path.join(ROOT, "highperformancengs_amd", "bin"), os.path.join(ROOT, "oracle", "_ref")
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
shards, per, rl = 8, int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 150
members = int(sys.argv[2]) if len(sys.argv) > 2 else 40  # 1e5 records per member
td = tempfile.mkdtemp(prefix="hpn_e2e_")
gz = [os.path.join(td, f"s{i}.fq.gz") for i in range(shards)]
with ThreadPoolExecutor(shards) as ex:
    list(ex.map(lambda i: L.orc_synth_write_fastq(gz[i].encode(), 5, i * per, per, rl, rl, members), range(shards)))
print("compressed bytes per file:", os.path.getsize(gz[0]), "members:", members, flush=True)
exe = os.path.join(BIN, "fastq_count")
for nfile in (1, 8):
    files = gz[:nfile]
    for env in ({"HPN_NO_MGZ": "1"}, {"HPN_GZ_THREADS": "1"}, {"HPN_GZ_THREADS": "2"}, {"HPN_GZ_THREADS": "4"},
                {"HPN_GZ_THREADS": "8"}, {"HPN_GZ_THREADS": "16"}, {}):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-t", str(nfile), "-o", os.path.join(td, "o.txt")] + files, cwd=td, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, env={**os.environ, **env})
            best = min(best, time.perf_counter() - t0)
        print(f"{nfile} file(s) {best:6.3f} s  {nfile*per*rl/best/1e9:6.3f} Gbases/s  {env}", flush=True)
subprocess.run(["rm", "-rf", td])
