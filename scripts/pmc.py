#!/usr/bin/env python3
"""rocprofv3 PMC passes for one kernel:  scripts/pmc.py <kernel-substring> "C1 C2" "C3 C4" ... -- python3 script.py args
One rocprofv3 run per counter group (own run each: no trace options beside --pmc); prints the per-launch mean of every counter
for dispatches whose kernel name contains the substring.  Run on the GPU box from the repo root."""
import csv
import glob
import os
import shutil
import subprocess
import sys

sep = sys.argv.index("--")
needles, groups, cmd = sys.argv[1].split(","), sys.argv[2:sep], sys.argv[sep + 1:]   # several kernels: a,b,c (one pass serves all)
os.environ.setdefault("TMPDIR", "/tmp")
for gi, g in enumerate(groups):
    d = f"gpurun_out/pmc_{os.getpid()}_{gi}"
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--pmc", *g.split(), "-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tot, cnt = {}, {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            for needle in needles:
                if needle in row["Kernel_Name"]:
                    k = (needle, row["Counter_Name"])
                    tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
                    cnt[k] = cnt.get(k, 0) + 1
    for k in sorted(tot):
        print(f"{k[1]} {tot[k] / cnt[k]:.0f} per launch ({k[0]}, {cnt[k]} launches)", flush=True)
    shutil.rmtree(d, ignore_errors=True)
