#!/usr/bin/env python3
"""K1 (k_tally_scan) on 150 GB of quality bytes at several read lengths, i.e. with 4 / 8 / 16 GB of offsets beside them:
separates the rate of the byte pass from that of the offset pass (profiles/r01e/k1_pair_pass_ab.txt)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import highperformancengs_amd as hp
ctx = hp.Context(0)
total = 150_000_000_000
for L in (150, 300, 75):
    n = total // L
    dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(1, 0, n, L, dq, None, do)
    ctx.sync()
    ts = []
    for r in range(6):
        ctx.fastq_tally_dev(dq, do, n)
        ctx.fastq_tally_fetch()
        if r: ts.append(ctx.last_kernel_ms(0))
    alg = n * L + (n + 1) * 8
    med = statistics.median(ts)
    print(f"L={L:4d} n={n:.3e}  {med:7.3f} ms  bytes {alg/1e9:6.1f} GB  {alg/med/1e6:7.1f} GB/s  (offsets {8*n/1e9:.1f} GB)", flush=True)
    del dq, do
    torch.cuda.empty_cache()
