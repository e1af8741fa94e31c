#!/usr/bin/env python3
"""A/B sweep of k_tally_scan variants, interleaved rounds in ONE process (MI355X guide rule 24).
HPN_K1_VARIANT = unroll*100 + nt*10 + dyn, HPN_K1_WG_PER_CU = workgroups per CU.  Prints median/min ms."""
import os
os.environ.setdefault("HPN_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "highperformancengs_amd", "testhooks", "libhpngs.so"))   # (HPN_K1_* are test-hooks switches)
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

n, L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000, 150
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ctx = hp.Context(0)
dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_fastq_dev(1, 0, n, L, dq, None, do)
ctx.sync()
alg = n * L + (n + 1) * 8
wgs = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (4, 8, 16)
variants = [(v, wg) for v in (410, 411, 800, 801, 810, 811, 1610, 1611) for wg in wgs]
times = {v: [] for v in variants}
ref = None
for r in range(rounds + 1):
    for v in variants:
        os.environ["HPN_K1_VARIANT"], os.environ["HPN_K1_WG_PER_CU"] = str(v[0]), str(v[1])
        ctx.fastq_tally_dev(dq, do, n)
        res = ctx.fastq_tally_fetch()
        key = (res.total, res.q20, res.q30, int(res.seqlen[L]))
        ref = ref or key
        assert key == ref, (v, key, ref)
        if r:
            times[v].append(ctx.last_kernel_ms(0))
print(f"n={n} reads x {L} bp, {alg/1e9:.1f} GB algorithmic per launch, {rounds} rounds")
for v in sorted(variants, key=lambda v: statistics.median(times[v])):
    med, mn = statistics.median(times[v]), min(times[v])
    print(f"unroll={v[0]//100:2d} nt={v[0]//10%10} dyn={v[0]%10} wg/cu={v[1]:2d}  median {med:7.3f} ms  min {mn:7.3f} ms  {alg/med/1e6:7.1f} GB/s")
