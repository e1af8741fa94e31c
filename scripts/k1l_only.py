#!/usr/bin/env python3
"""K1L alone on 2e8 x 150 bp: `python scripts/k1l_only.py [reps] [q|qn]` (q: Quality matrix, qn: + Nucleotide) -> ms."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "q"
ctx = hp.Context(0)
n, L = 200_000_000, int(os.environ.get("READ_LEN", 150))
dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
db = torch.empty(n * L, dtype=torch.uint8, device="cuda") if mode == "qn" else None
do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_fastq_dev(7, 0, n, L, dq, db, do)
ctx.sync()
ts = []
for r in range(reps + 1):
    ctx.fastq_tally_dev(dq, do, n, d_base=db, flags=3 if mode == "qn" else 1)
    res = ctx.fastq_tally_fetch(qual_hist=True, nuc_hist=(mode == "qn"))
    if r:
        ts.append(ctx.last_kernel_ms(0))
assert os.environ.get("DIAG") or res.total == n * L
ms = statistics.median(ts)
byts = (2 if mode == "qn" else 1) * n * L + 8 * (n + 1)
print(f"{os.environ.get('HPN_LIB', 'default'):28s} K1L[{mode}] {ms:.3f} ms  {byts / ms / 1e6:.0f} GB/s  frac {byts / ms / 1e6 / 8000:.3f}")
