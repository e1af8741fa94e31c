#!/usr/bin/env python3
"""Same-session A/B of the bam2depth kernels (K3 = hpn_depth_add_dev, K4 = hpn_depth_finish) on the chr1-at-30x shape.
Run once per variant library:  HPN_LIB=.scratch/ab/<name>/libhpngs.so python scripts/ab_depth.py [reps]
Prints one line: variant, K3 ms, K4 ms, K5 ms (medians).  No result checks: diagnostic builds leave work out."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

ctx = hp.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
# ---- BAM: chr1-sized target at 30x -----------------------------------------------------------
if True:
    TL, L = 248_956_422, 150
    n = 30 * TL // L
    g = torch.Generator(device="cuda").manual_seed(5)
    pos = torch.sort(torch.randint(0, TL - L, (n,), device="cuda", generator=g, dtype=torch.int32)).values
    tid = torch.zeros(n, dtype=torch.int32, device="cuda")
    fl = torch.tensor([0, 16, 0, 16, 0, 16, 0, 16, 0, 16, 0, 16, 0, 16, 0, 16, 0, 16, 4, 256, 512, 1024], dtype=torch.int32,
                      device="cuda")[torch.randint(0, 22, (n,), device="cuda", generator=g)]
    # CIGAR mix of SURVEY §8d: 85 % 150M, 5 % 40M2I108M, 5 % 60M5D90M, 5 % 10S140M
    pick = torch.randint(0, 20, (n,), device="cuda", generator=g)
    kind = torch.where(pick < 17, 0, pick - 16)
    table = torch.tensor([[150 << 4, 0, 0], [40 << 4, (2 << 4) | 1, 108 << 4], [60 << 4, (5 << 4) | 2, 90 << 4],
                          [(10 << 4) | 4, 140 << 4, 0]], dtype=torch.int32, device="cuda")
    ncig = torch.tensor([1, 3, 3, 2], dtype=torch.int32, device="cuda")[kind]
    words = table[kind]
    keep = torch.arange(3, device="cuda")[None, :] < ncig[:, None]
    cigar = words[keep].contiguous()
    cigar_off = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    cigar_off[1:] = torch.cumsum(ncig, 0)
    lq = torch.full((n,), L, dtype=torch.int32, device="cuda")
    seq_off = torch.arange(n + 1, device="cuda", dtype=torch.int64) * ((L + 1) // 2)
    seq4 = torch.randint(0, 256, (n * ((L + 1) // 2),), device="cuda", generator=g, dtype=torch.uint8)

    class D:
        pass
    d = D()
    d.tid, d.pos, d.flag, d.l_qseq, d.cigar_off, d.cigar, d.seq_off, d.seq4 = tid, pos, fl, lq, cigar_off, cigar, seq_off, seq4
    n_ops = int(cigar.numel())
    n_m = int(((cigar & 15) == 0).sum().item())
    keep_alive = []
    keep_alive = []
    ts3, ts4 = [], []
    for r in range(reps + 1):
        ctx._ck(ctx.L.hpn_depth_begin(ctx.h, 0, TL, 0x704), "begin")
        b = ctx._batch(d, keep_alive)
        ctx._ck(ctx.L.hpn_depth_add_dev(ctx.h, C.byref(b)), "add")
        ctx.sync()
        t3 = ctx.last_kernel_ms(2)
        runs, win = ctx.depth_finish(TL, 20000, runs_cap=1 << 27)
        t4 = ctx.last_kernel_ms(2)
        if r:
            ts3.append(t3), ts4.append(t4)
    tsf = []
    for r in range(reps + 1):
        text_bytes = ctx.depth_bedgraph_format("chr1")
        if r:
            tsf.append(ctx.last_kernel_ms(2))
    off = np.array([0, TL // 20000 + 1], np.uint64)
    ts5 = []
    for r in range(reps + 1):
        bins, gc, ln, touched, nc = ctx.window_counts(d, off, 20000, dev=True)
        if r:
            ts5.append(ctx.last_kernel_ms(3))
    print("%-28s K3 %.3f ms   K4 %.3f ms   text %.3f ms   K5 %.3f ms   (runs %d, gc %d)" % (os.environ.get("HPN_LIB", "default"), statistics.median(ts3),
                                                        statistics.median(ts4), statistics.median(tsf), statistics.median(ts5), len(runs), int(gc.sum())), flush=True)
