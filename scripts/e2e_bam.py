#!/usr/bin/env python3
"""End-to-end (file -> reports) of the BAM tools vs the reference binaries on one BAM
(default .scratch/big.bam: 2 M x 150M reads over 2 x 5 Mb, the shape of SURVEY A.5)."""
import filecmp
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, REF = os.path.join(ROOT, "highperformancengs_amd", "bin"), os.path.join(ROOT, "oracle", "_ref")
bam = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, ".scratch", "big.bam"))
name = os.path.basename(bam)


def run(exe, args, cwd):
    t0 = time.perf_counter()
    p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return time.perf_counter() - t0, p


out = {}
for tool, args in (("bam2depth", ["-o", "d", name]), ("bam2wig", ["-o", "w", name]), ("bam_sliding_count", ["-o", "s", name])):
    for who, d in (("reference", REF), ("hpngs", BIN)):
        exe = os.path.join(d, tool)
        if not os.access(exe, os.X_OK):
            print(f"{tool:18s} {who:9s} (binary not available)")
            continue
        td = tempfile.mkdtemp(prefix="hpn_bam_")
        os.symlink(bam, os.path.join(td, name))
        os.symlink(bam + ".bai", os.path.join(td, name + ".bai"))
        run(exe, args, td) if who == "hpngs" else None  # warm the GPU runtime / page cache
        for f in os.listdir(td):
            if not f.startswith(name):
                os.unlink(os.path.join(td, f))
        dt, p = run(exe, args, td)
        files = sorted(f for f in os.listdir(td) if os.path.isfile(os.path.join(td, f)) and not os.path.islink(os.path.join(td, f)))
        out[(tool, who)] = (td, files)
        print(f"{tool:18s} {who:9s} {dt:7.3f} s  rc={p.returncode}  files={files}")
    a, b = out.get((tool, "reference")), out.get((tool, "hpngs"))
    if a and b:
        same = a[1] == b[1] and all(filecmp.cmp(os.path.join(a[0], f), os.path.join(b[0], f), shallow=False) for f in a[1])
        print(f"   outputs identical: {same}")
for td, _ in out.values():
    shutil.rmtree(td, ignore_errors=True)
