#!/usr/bin/env python3
"""fastq_trim on a bgzip-compressed FASTQ: GPU inflate + device framing/cut vs host BGZF threads vs the reference."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, REF = os.path.join(ROOT, "highperformancengs_amd", "bin"), os.path.join(ROOT, "oracle", "_ref")
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
reads, rl = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 150
td = tempfile.mkdtemp(prefix="hpn_e2e_")
plain, bg = os.path.join(td, "p.fq"), os.path.join(td, "b.fq.gz")
L.orc_synth_write_fastq(plain.encode(), 5, 0, reads, rl, rl, 0)
subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "scripts", "bam_synth.cpp"), "-o", os.path.join(td, "bs"), "-lz", "-lpthread"])
subprocess.check_call([os.path.join(td, "bs"), "--bgzip", plain, bg, "16"])
os.unlink(plain)
print(f"{reads} reads x {rl}, {os.path.getsize(bg)/1e6:.0f} MB bgzip", flush=True)
outs = []
for who, exe, env in (("reference", os.path.join(REF, "fastq_trim"), {}), ("hpngs GPU inflate", os.path.join(BIN, "fastq_trim"), {}),
                      ("hpngs host BGZF threads", os.path.join(BIN, "fastq_trim"), {"HPN_BAM_GPU": "0"})):
    if not os.access(exe, os.X_OK):
        continue
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        subprocess.run([exe, "-i", bg, "-o", "t", "-s", "5", "-e", "120"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, **env}, check=True)
        best = min(best, time.perf_counter() - t0)
    outs.append(subprocess.run("md5sum < t.trim.fastq", shell=True, cwd=td, stdout=subprocess.PIPE).stdout)
    print(f"fastq_trim -s 5 -e 120  {who:26s} {best:7.3f} s  {reads*rl/best/1e9:6.3f} Gbases/s", flush=True)
print("   outputs identical:", len(set(outs)) == 1)
subprocess.run(["rm", "-rf", td])
