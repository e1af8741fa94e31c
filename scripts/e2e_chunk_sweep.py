#!/usr/bin/env python3
"""fastq_count on 8 plain files x 4e6 reads: wall time vs text chunk size / reader threads (HPN_TIMING lines of one run)."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")   # (the build that reads test / timing switches)
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
shards, per, rl = 8, int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 150
td = tempfile.mkdtemp(prefix="hpn_e2e_")
plain = [os.path.join(td, f"s{i}.fq") for i in range(shards)]
with ThreadPoolExecutor(shards) as ex:
    list(ex.map(lambda i: L.orc_synth_write_fastq(plain[i].encode(), 5, i * per, per, rl, rl, 0), range(shards)))
exe = os.path.join(BIN, "fastq_count")
for env in ({}, {"HPN_TEXT_CHUNK": str(4 << 20)}, {"HPN_TEXT_CHUNK": str(8 << 20)}, {"HPN_TEXT_CHUNK": str(16 << 20)},
            {"HPN_TEXT_CHUNK": str(8 << 20), "HPN_READ_THREADS": "2"}, {"HPN_TEXT_CHUNK": str(8 << 20), "HPN_READ_THREADS": "1"}):
    best, err = 1e9, b""
    for _ in range(3):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-t", "8", "-o", os.path.join(td, "o.txt")] + plain, cwd=td, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1", **env})
        dt = time.perf_counter() - t0
        if dt < best:
            best, err = dt, p.stderr
    print(f"{best:6.3f} s  {shards*per*rl/best/1e9:6.2f} Gbases/s  {env}")
    if not env:
        print(err.decode())
subprocess.run(["rm", "-rf", td])
