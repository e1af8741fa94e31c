#!/usr/bin/env python3
"""Fixed per-process cost of the tools: library load, HIP init, context, teardown."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")   # (the build that reads test / timing switches)
open("/tmp/empty.fq", "w").close()
open("/tmp/one.fq", "w").write("@a\nACGT\n+\nIIII\n")


def t(cmd, env=None, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, **(env or {})})
        best = min(best, time.perf_counter() - t0)
    print(f"{best*1e3:8.1f} ms  {' '.join(os.path.basename(c) for c in cmd)}  {env or ''}  | {p.stderr.decode().strip().splitlines()[-1:]}", flush=True)


t(["/bin/true"])
t([os.path.join(B, "fastq_count"), "-h"])
t([os.path.join(B, "fastq_count"), "/tmp/empty.fq"])
t([os.path.join(B, "fastq_count"), "/tmp/one.fq"], {"HPN_TIMING": "1"})
t([os.path.join(B, "fastq_count"), "/tmp/one.fq"], {"HPN_TEXT": "0"})
t([os.path.join(B, "fastq_count"), "/tmp/one.fq"], {"HPN_TEXT_CHUNK": "65536"})
t([os.path.join(B, "fastq_trim"), "-i", "/tmp/one.fq", "-o", "/tmp/x"])
t([os.path.join(ROOT, "oracle", "_ref", "fastq_count"), "/tmp/one.fq"])
