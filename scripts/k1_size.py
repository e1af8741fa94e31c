#!/usr/bin/env python3
"""k_tally_scan rate vs launch size and vs position inside one large allocation."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

n, L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000, 150
ctx = hp.Context(0)
dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_fastq_dev(1, 0, n, L, dq, None, do)
ctx.sync()


def run(first, cnt, reps=7):
    ts = []
    for r in range(reps + 1):
        ctx.fastq_tally_dev(dq, do[first:], cnt)
        res = ctx.fastq_tally_fetch()
        assert res.total == cnt * L
        if r:
            ts.append(ctx.last_kernel_ms(0))
    return statistics.median(ts)


for first, cnt in [(0, n // 10), (0, n // 5), (0, 2 * n // 5), (3 * n // 5, 2 * n // 5), (0, 3 * n // 5), (0, 4 * n // 5),
                   (0, n), (n // 10, n // 10), (9 * n // 10, n // 10), (0, n)]:
    ms = run(first, cnt)
    alg = cnt * L + (cnt + 1) * 8
    print(f"records [{first:>11d}, +{cnt:>11d})  {ms:8.3f} ms  {alg/ms/1e6:7.1f} GB/s")
