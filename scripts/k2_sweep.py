#!/usr/bin/env python3
"""K2 (k_trim_scan + k_trim_copy, -s 5 -e 140) vs workgroups per CU of the copy kernel (HPN_TRIM_WG_PER_CU), interleaved
rounds in one process.  Prints median ms of the two kernels together and the algorithmic rate."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

n, L, S, E = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000, 150, 5, 140
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wgs = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 3, 4, 6, 8, 12, 16)
ctx = hp.Context(0)
dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_fastq_dev(7, 0, n, L, dq, db, do)
oq = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
ob = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.sync()
alg = 2 * n * L + 16 * n + 2 * n * (E - S) + 8 * n
times = {w: [] for w in wgs}
ref = None
for r in range(rounds + 1):
    for w in wgs:
        os.environ["HPN_TRIM_WG_PER_CU"] = str(w)
        ctx.fastq_trim_dev(db, dq, do, n, S, E, ob, oq, oo)
        ctx.sync()
        key = (int(oo[-1].item()), int(oq[:: 1 << 20].to(torch.int64).sum().item()))
        ref = ref or key
        assert key == ref, (w, key, ref)
        if r:
            times[w].append(ctx.last_kernel_ms(1))
print(f"n={n} reads x {L} bp, {alg/1e9:.1f} GB algorithmic per call, {rounds} rounds")
for w in sorted(wgs, key=lambda w: statistics.median(times[w])):
    med = statistics.median(times[w])
    print(f"copy wg/cu={w:2d}  median {med:7.3f} ms  min {min(times[w]):7.3f} ms  {alg/med/1e6:7.1f} GB/s")
