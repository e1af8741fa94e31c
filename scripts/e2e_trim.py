#!/usr/bin/env python3
"""fastq_trim end to end on one larger plain file: reference vs fast path vs host framer, outputs compared."""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, REF = os.path.join(ROOT, "highperformancengs_amd", "bin"), os.path.join(ROOT, "oracle", "_ref")
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
per, rl = int(float(sys.argv[1])), 150
td = tempfile.mkdtemp(prefix="hpn_e2e_")
f = os.path.join(td, "s.fq")
L.orc_synth_write_fastq(f.encode(), 5, 0, per, rl, rl, 0)


def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as fh:
        while True:
            b = fh.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


sums = {}
for who, d, env in (("reference", REF, {}), ("hpngs", BIN, {"HPN_TIMING": "1"}), ("hpngs-host", BIN, {"HPN_TEXT": "0"})):
    exe = os.path.join(d, "fastq_trim")
    if not os.access(exe, os.X_OK):
        continue
    for rep in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-i", f, "-s", "5", "-e", "140", "-o", os.path.join(td, who)], cwd=td, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env={**os.environ, **env})
        dt = time.perf_counter() - t0
    print(f"fastq_trim  {who:11s} 1 x {per} reads: {dt:7.3f} s  {per*rl/dt/1e9:7.3f} Gbases/s", flush=True)
    if env.get("HPN_TIMING"):
        print(p.stderr.decode().strip())
    sums[who] = md5(os.path.join(td, who + ".trim.fastq"))
print("outputs identical:", len(set(sums.values())) == 1, sums)
subprocess.run(["rm", "-rf", td])
