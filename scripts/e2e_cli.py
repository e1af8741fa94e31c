#!/usr/bin/env python3
"""End-to-end (file -> report) rate of the drop-in tools vs the reference binaries on the same
files: host inflate / framing / PCIe included.  Never the bench `value`; see DESIGN.md §5."""
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
shards, per, rl = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000, 150
td = tempfile.mkdtemp(prefix="hpn_e2e_")
plain = [os.path.join(td, f"s{i}.fq") for i in range(shards)]
gz = [p + ".gz" for p in plain]
with ThreadPoolExecutor(shards) as ex:
    list(ex.map(lambda i: L.orc_synth_write_fastq(plain[i].encode(), 5, i * per, per, rl, rl, 0), range(shards)))
    list(ex.map(lambda i: L.orc_synth_write_fastq(gz[i].encode(), 5, i * per, per, rl, rl, 16), range(shards)))
bases = shards * per * rl


def run(cmd, cwd=td):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return time.perf_counter() - t0, p


for label, files in (("plain", plain), ("gzip", gz)):
    for tool in ("fastq_count", "fastq_count_kthread"):
        outs = {}
        for who, d in (("reference", REF), ("hpngs", BIN)):
            exe = os.path.join(d, tool)
            if not os.access(exe, os.X_OK):
                continue
            run([exe, "-t", str(shards), "-o", os.path.join(td, "o.txt")] + files)  # warm page cache / GPU
            dt, p = run([exe, "-t", str(shards), "-o", os.path.join(td, f"{who}.txt")] + files)
            outs[who] = sorted(open(os.path.join(td, f"{who}.txt")).read().splitlines())
            print(f"{tool:22s} {label:5s} {who:9s} {shards} files x {per} reads: {dt:7.3f} s  {bases/dt/1e9:7.3f} Gbases/s  rc={p.returncode}")
        if len(outs) == 2:
            print("   reports identical:", outs["reference"] == outs["hpngs"])
    outs = {}
    for who, d in (("reference", REF), ("hpngs", BIN)):  # one stream: no file-level parallelism to hide behind
        exe = os.path.join(d, "fastq_count")
        if os.access(exe, os.X_OK):
            dt, p = run([exe, "-o", os.path.join(td, f"{who}.1.txt"), files[0]])
            outs[who] = open(os.path.join(td, f"{who}.1.txt")).read()
            print(f"{'fastq_count':22s} {label:5s} {who:9s} 1 file x {per} reads: {dt:7.3f} s  {per*rl/dt/1e9:7.3f} Gbases/s  rc={p.returncode}")
    if len(outs) == 2:
        print("   reports identical:", outs["reference"] == outs["hpngs"])
    if label == "plain":
        for who, d in (("reference", REF), ("hpngs", BIN)):
            exe = os.path.join(d, "fastq_trim")
            if os.access(exe, os.X_OK):
                dt, p = run([exe, "-i", files[0], "-s", "5", "-e", "140", "-o", os.path.join(td, who)])
                print(f"{'fastq_trim':22s} {label:5s} {who:9s} 1 file x {per} reads: {dt:7.3f} s  {per*rl/dt/1e9:7.3f} Gbases/s")
        a, b = (open(os.path.join(td, w + ".trim.fastq"), "rb").read() for w in ("reference", "hpngs")) if all(
            os.path.exists(os.path.join(td, w + ".trim.fastq")) for w in ("reference", "hpngs")) else (b"", b"x")
        print("   trim outputs identical:", a == b)
subprocess.run(["rm", "-rf", td])
