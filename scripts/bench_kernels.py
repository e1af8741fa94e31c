#!/usr/bin/env python3
"""Per-kernel rates of the non-headline kernels (K1L, K2, K3, K4, K5) on BASELINE-shaped
device-resident inputs.  One JSON line per kernel: ms, algorithmic GB/s, fraction of 8 TB/s.
Algorithmic bytes follow SURVEY.md §8(d) / DESIGN.md §3."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

PEAK = 8000.0
ctx = hp.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None


def emit(name, ms, alg_bytes, **extra):
    gbs = alg_bytes / ms / 1e6
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes), "GBps": round(gbs, 1),
                      "frac_of_8TBps": round(gbs / PEAK, 4), **extra}), flush=True)


def med(fn, fam):
    ts = []
    for r in range(reps + 1):
        fn()
        ctx.sync()
        if r:
            ts.append(ctx.last_kernel_ms(fam))
    return statistics.median(ts)


# ---- FASTQ: K1L and K2 on 2e8 x 150 bp ---------------------------------------------------
if not only or only & {"k1l", "k2"}:
    n, L = 200_000_000, 150
    dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(7, 0, n, L, dq, db, do)
    ctx.sync()
    if not only or "k1l" in only:
        def k1l():
            ctx.fastq_tally_dev(dq, do, n, flags=1)
            ctx.fastq_tally_fetch(qual_hist=True)
        emit("K1L k_tally_hist (Quality[128][512])", med(k1l, 0), n * L + (n + 1) * 8, reads=n, read_len=L)

        def k1ln():
            ctx.fastq_tally_dev(dq, do, n, d_base=db, flags=3)
            ctx.fastq_tally_fetch(qual_hist=True, nuc_hist=True)
        emit("K1L k_tally_hist (+Nucleotide[5][512])", med(k1ln, 0), 2 * n * L + (n + 1) * 8, reads=n, read_len=L)
    if not only or "k2" in only:
        S, E = 5, 140
        oq = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
        ob = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
        oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        alg = 2 * n * L + 16 * n + 2 * n * (E - S) + 8 * n
        emit("K2 k_trim_scan+k_trim_copy (-s 5 -e 140)", med(lambda: ctx.fastq_trim_dev(db, dq, do, n, S, E, ob, oq, oo), 1),
             alg, reads=n, read_len=L)
        del oq, ob, oo
    del dq, db, do
    torch.cuda.empty_cache()

# ---- BAM: chr1-sized target at 30x -----------------------------------------------------------
if not only or only & {"k3", "k4", "k5"}:
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
    if not only or only & {"k3", "k4"}:
        import ctypes as C
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
        slots = TL + 1 + (1 << 21)
        # sorted records take the swept route (bam_depth.hip): add = k_depth_index + k_depth_sweep (difference array, prefix sum and
        # runs in one pass, nothing through HBM), finish = k_depth_scan over the last tiles only -- so the two are ONE leg; the
        # two-pass route's K3 and K4 by themselves are timed by scripts/bench_depth_legs.py
        alg = n * 16 + 4 * n_ops + 8 * n_m + slots * 4 + 12 * len(runs) + 8 * len(win)
        emit("K3+K4 swept route: k_depth_index + k_depth_sweep (+ k_depth_scan over the last tiles)", statistics.median(ts3) + statistics.median(ts4),
             alg, add_ms=round(statistics.median(ts3), 4), finish_ms=round(statistics.median(ts4), 4), records=n, cigar_ops=n_ops,
             positions=slots, runs=len(runs))
        mean_cov = float(win.sum()) / TL
        assert 20 < mean_cov < 31, mean_cov
    if not only or "k5" in only:
        off = np.array([0, TL // 20000 + 1], np.uint64)
        ts5 = []
        for r in range(reps + 1):
            bins, gc, ln, touched, nc = ctx.window_counts(d, off, 20000, dev=True)
            if r:
                ts5.append(ctx.last_kernel_ms(3))
        assert int(bins.sum()) == nc
        emit("K5 k_window_add", statistics.median(ts5), n * (20 + (L + 1) // 2), records=n)
