"""One chr1-sized target at 30x through the sweep (hpn_depth_begin_w / add_dev / finish), and once through the two-pass route:
the smallest program that runs k_depth_index / k_depth_sweep / k_depth_tiles / k_depth_scan at full size (for rocprofv3)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import highperformancengs_amd as hp

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = hp.Context(0)
TL, L = 248_956_422, 150
n = 30 * TL // L
g = torch.Generator(device="cuda").manual_seed(5)
pos = torch.sort(torch.randint(0, TL - 200, (n,), device="cuda", generator=g, dtype=torch.int32)).values
tid = torch.zeros(n, dtype=torch.int32, device="cuda")
fl = torch.tensor([0, 16] * 9 + [4, 256, 512, 1024], dtype=torch.int32, device="cuda")[torch.randint(0, 22, (n,), device="cuda", generator=g)]
pick = torch.randint(0, 20, (n,), device="cuda", generator=g)
kind = torch.where(pick < 17, 0, pick - 16)
table = torch.tensor([[150 << 4, 0, 0], [40 << 4, (2 << 4) | 1, 108 << 4], [60 << 4, (5 << 4) | 2, 90 << 4], [(10 << 4) | 4, 140 << 4, 0]], dtype=torch.int32, device="cuda")
ncig = torch.tensor([1, 3, 3, 2], dtype=torch.int32, device="cuda")[kind]
words = table[kind]
keep = torch.arange(3, device="cuda")[None, :] < ncig[:, None]
cigar = words[keep].contiguous()
cigar_off = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
cigar_off[1:] = torch.cumsum(ncig, 0)


class D:
    pass


d = D()
d.tid, d.pos, d.flag, d.cigar_off, d.cigar = tid, pos, fl, cigar_off, cigar
d.l_qseq = torch.full((n,), L, dtype=torch.int32, device="cuda")
d.seq_off, d.seq4 = None, None
keep_alive = []
for mode in ("sweep", "two-pass"):
    for r in range(reps + 1):
        if mode == "sweep":
            ctx._ck(ctx.L.hpn_depth_begin_w(ctx.h, 0, TL, 0x704, 20000), "begin")
        else:
            ctx._ck(ctx.L.hpn_depth_begin(ctx.h, 0, TL, 0x704 | 0x80000000), "begin")
        b = ctx._batch(d, keep_alive)
        ctx._ck(ctx.L.hpn_depth_add_dev(ctx.h, C.byref(b)), "add")
        ctx.sync()
        t_add = ctx.last_kernel_ms(2)
        nr = C.c_uint64(0)
        ctx._ck(ctx.L.hpn_depth_finish(ctx.h, 20000, None, 0, C.byref(nr), None), "finish")
        t_fin = ctx.last_kernel_ms(2)
    print(f"{mode}: add {t_add:.3f} ms  finish {t_fin:.3f} ms  runs {nr.value}")
    if mode == "sweep":
        ts = []
        for r in range(reps + 1):
            ctx.depth_bedgraph_format("chr1")
            ts.append(ctx.last_kernel_ms(2))
        print(f"bedgraph text: {min(ts):.3f} ms")
