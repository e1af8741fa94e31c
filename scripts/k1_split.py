#!/usr/bin/env python3
"""Does one long grid-stride launch lose rate to workgroup drift?  Same 1e9-read pass as 1..K launches."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

n, L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000, 150
ctx = hp.Context(0)
dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_fastq_dev(1, 0, n, L, dq, None, do)
ctx.sync()
alg = n * L + (n + 1) * 8
for parts in (1, 2, 4, 5, 8, 10, 16, 32, 64, 1):
    ts = []
    for r in range(6):
        ctx.sync()
        t0 = time.perf_counter()
        for p in range(parts):
            a, b = p * n // parts, (p + 1) * n // parts
            ctx.fastq_tally_dev(dq, do[a:], b - a)
        res = ctx.fastq_tally_fetch()
        dt = (time.perf_counter() - t0) * 1e3
        assert res.total == n * L and int(res.seqlen[L]) == n
        if r:
            ts.append(dt)
    ms = statistics.median(ts)
    print(f"{parts:3d} launches: {ms:8.3f} ms wall  {alg/ms/1e6:7.1f} GB/s")
