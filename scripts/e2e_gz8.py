#!/usr/bin/env python3
"""8 single-member .fastq.gz files, fastq_count -t 8 (the kthread use case): GPU two-pass inflate vs host readers vs reference."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, REF = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")   # (the build that reads test / timing switches), os.path.join(ROOT, "oracle", "_ref")
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
shards, per, rl = 8, int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 150
td = tempfile.mkdtemp(prefix="hpn_e2e_")
gz = [os.path.join(td, f"s{i}.fq.gz") for i in range(shards)]
with ThreadPoolExecutor(shards) as ex:
    list(ex.map(lambda i: L.orc_synth_write_fastq(gz[i].encode(), 5, i * per, per, rl, rl, 1), range(shards)))
print(f"{shards} files x {per} reads x {rl}, {os.path.getsize(gz[0])/1e6:.0f} MB each, one gzip member each", flush=True)
outs = []
for who, exe, env in (("reference", os.path.join(REF, "fastq_count"), {}), ("hpngs GPU two-pass inflate", os.path.join(BIN, "fastq_count"), {}),
                      ("hpngs host readers", os.path.join(BIN, "fastq_count"), {"HPN_GZ_GPU": "0"}),
                      ("hpngs zlib", os.path.join(BIN, "fastq_count"), {"HPN_GZ_GPU": "0", "HPN_NO_MGZ": "1"})):
    if not os.access(exe, os.X_OK):
        continue
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-t", "8"] + gz, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, **env})
        best = min(best, time.perf_counter() - t0)
    outs.append(b"\n".join(sorted(p.stdout.split(b"\n"))))
    print(f"fastq_count -t 8, 8 files  {who:28s} {best:7.3f} s  {shards*per*rl/best/1e9:6.3f} Gbases/s", flush=True)
print("   reports identical (rows sorted):", len(set(outs)) == 1)
p = subprocess.run([os.path.join(BIN, "fastq_count"), "-t", "8"] + gz, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                   env={**os.environ, "HPN_TIMING": "1", "HPN_GZ_DEBUG": "1"})
print(p.stderr.decode())
for i in range(4):
    p = subprocess.run([os.path.join(BIN, "fastq_count"), gz[0]], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_TIMING": "1", "HPN_GZ_DEBUG": "1"})
    print("\n".join(l for l in p.stderr.decode().splitlines() if "gz" in l))
subprocess.run(["rm", "-rf", td])
