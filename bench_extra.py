"""bench_extra.py -- the other named configurations, reported beside bench.py's headline in its `extra` object.

  kernel_legs(ctx)      K1L (fastq_count_kthread -L), K2 (fastq_trim, BASELINE configs[2] shape), K3 + K4 (bam2depth on a
                        chr1-sized target at 30x, configs[3]) and K5 (bam_sliding_count): device-resident synthetic input,
                        algorithmic bytes of SURVEY.md §8(d), kernel time from the library's HIP events -> fraction of 8 TB/s.
  exact_check(...)      the headline kernel's counts on the full resident batch against an independent kernel (K1L's
                        Quality matrix: total / Q20 / Q30 are its row sums) and, after all timing, three windows of the
                        batch against the CPU oracle (checker use only).
  e2e_legs(...)         file -> report through the built CLI binaries (host inflate / framing / PCIe / text output included),
                        the reference binary timed beside each when oracle/_ref holds it.  Never the headline `value`.
"""
import ctypes as C
import filecmp
import json
import os
import shutil
import statistics
import subprocess
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")
PEAK = 8000.0


def _leg(name, ms, alg_bytes, bytes_touched=None, **more):
    """algorithmic_bytes: SURVEY.md 8(d)'s formula for the kernel, the figure `frac` is computed from.  bytes_touched: what
    the kernel really has to move where that differs (wider offsets, fields 8(d) does not count) -- reported, not priced."""
    gbs = alg_bytes / ms / 1e6
    leg = {"kernel": name, "kernel_ms": round(ms, 4), "algorithmic_bytes": int(alg_bytes), "achieved_GBps": round(gbs, 1),
           "frac": round(gbs / PEAK, 4), **more}
    if bytes_touched is not None:
        leg["bytes_touched"] = int(bytes_touched)
        leg["frac_bytes_touched"] = round(bytes_touched / ms / 1e6 / PEAK, 4)
    return leg


def _median_ms(ctx, fn, family, reps):
    ts = []
    for r in range(reps + 1):
        fn()
        ctx.sync()
        if r:
            ts.append(ctx.last_kernel_ms(family))
    return statistics.median(ts)


# --------------------------------------------------------------------------------------------------------------
# kernel legs
# --------------------------------------------------------------------------------------------------------------
def plain_traffic():
    """What the part gives to plain traffic on this box (torch, 8 GiB): a device-to-device copy (read + write counted) and a
    fill.  Kernels that write about as much as they read (K2, K3, K4) are priced against 8 TB/s like the others; this is the
    rate a copy reaches beside them."""
    import torch
    x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    x.fill_(3), y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    for name, fn, nbytes in (("copy_read_plus_write", lambda: y.copy_(x), 2 * x.numel()), ("fill_write", lambda: y.fill_(7), x.numel())):
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[name + "_GBps"] = round(nbytes / ms / 1e6, 1)
    del x, y
    torch.cuda.empty_cache()
    return out


def kernel_legs(ctx, reps=5, fastq=True):
    import torch
    legs = []
    if fastq:
        _fastq_kernel_legs(ctx, reps, legs)
    _bam_kernel_legs(ctx, reps, legs)
    return legs


def _fastq_kernel_legs(ctx, reps, legs):
    import torch
    # ---- FASTQ: K1L and K2 on 2e8 x 150 bp ------------------------------------------------------------------
    n, L = 200_000_000, 150
    dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(7, 0, n, L, dq, db, do)
    ctx.sync()

    def k1l():
        ctx.fastq_tally_dev(dq, do, n, flags=1)
        ctx.fastq_tally_fetch(qual_hist=True)
    legs.append(_leg("K1L k_tally_hist: Quality[128][512] (fastq_count_kthread -L)", _median_ms(ctx, k1l, 0, reps),
                     n * L + (n + 1) * 8, reads=n, read_len=L))

    def k1ln():
        ctx.fastq_tally_dev(dq, do, n, d_base=db, flags=3)
        ctx.fastq_tally_fetch(qual_hist=True, nuc_hist=True)
    legs.append(_leg("K1L k_tally_hist: + Nucleotide[5][512] (hpn_fastq_rqc)", _median_ms(ctx, k1ln, 0, reps),
                     2 * n * L + (n + 1) * 8, reads=n, read_len=L))
    # K1 on reads of mixed lengths (trimmed data): the offset pass cannot take its one-add-per-1024-equal-lengths shortcut
    g0 = torch.Generator(device="cuda").manual_seed(11)
    rl = torch.randint(100, 152, (n,), device="cuda", generator=g0, dtype=torch.int64)
    ro = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(rl, 0, out=ro[1:])
    tot = int(ro[-1].item())
    assert tot <= n * L                               # lengths <= 151 on average 125.5: the bytes of dq[:tot] serve

    def k1r():
        ctx.fastq_tally_dev(dq, ro, n, flags=0)
        return ctx.fastq_tally_fetch()
    ms = _median_ms(ctx, k1r, 0, reps)
    fast = k1r()
    ctx.fastq_tally_dev(dq, ro, n, flags=1)
    full = ctx.fastq_tally_fetch(qual_hist=True)      # exact: the independent kernel over the same ragged batch
    rows = np.asarray(full.qual_hist, np.uint64).sum(axis=1)
    assert (fast.total, fast.q20, fast.q30) == (tot, int(rows[53:].sum()), int(rows[63:].sum())) and int(rows.sum()) == tot
    assert np.array_equal(np.asarray(fast.seqlen), np.asarray(full.seqlen)) and int(fast.seqlen[100:152].sum()) == n
    legs.insert(0, _leg("K1 k_tally_scan on ragged reads (lengths 100..151)", ms, tot + (n + 1) * 8, reads=n, bases=tot))
    del rl, ro
    S, E = 5, 140
    oq = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
    ob = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
    oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ms = _median_ms(ctx, lambda: ctx.fastq_trim_dev(db, dq, do, n, S, E, ob, oq, oo), 1, reps)
    # 8(d): read 2*Sum(len) + 8n, write 2*Sum(newlen) + 8n; the kernels read the 8-byte offsets twice (scan, then copy)
    legs.append(_leg("K2 k_trim_scan + k_trim_copy: fastq_trim -s 5 -e 140, one mate", ms,
                     2 * n * L + 8 * n + 2 * n * (E - S) + 8 * n, bytes_touched=2 * n * L + 16 * n + 2 * n * (E - S) + 8 * n,
                     reads=n, read_len=L))
    # trimmed output = strided view of the input (exact, on the device)
    assert int(oo[-1].item()) == n * (E - S)
    k = 1_000_003
    assert torch.equal(oq[k * (E - S):(k + 1) * (E - S)], dq[k * L + S:k * L + E])
    del dq, db, do, oq, ob, oo
    torch.cuda.empty_cache()


def _bam_kernel_legs(ctx, reps, legs, raw_only=False):
    """raw_only: the raw-record legs alone, nothing compared with the SoA route (PMC passes: scripts/bench_raw_legs.py raw)"""
    import torch
    # ---- BAM: chr1-sized target at 30x (SURVEY §8d CIGAR / flag mix) ------------------------------------------
    TL, L = 248_956_422, 150
    n = 30 * TL // L
    g = torch.Generator(device="cuda").manual_seed(5)
    pos = torch.sort(torch.randint(0, TL - 200, (n,), device="cuda", generator=g, dtype=torch.int32)).values
    tid = torch.zeros(n, dtype=torch.int32, device="cuda")
    fl = torch.tensor([0, 16] * 9 + [4, 256, 512, 1024], dtype=torch.int32, device="cuda")[
        torch.randint(0, 22, (n,), device="cuda", generator=g)]
    pick = torch.randint(0, 20, (n,), device="cuda", generator=g)
    kind = torch.where(pick < 17, 0, pick - 16)       # 85 % 150M, 5 % 40M2I108M, 5 % 60M5D90M, 5 % 10S140M
    table = torch.tensor([[150 << 4, 0, 0], [40 << 4, (2 << 4) | 1, 108 << 4], [60 << 4, (5 << 4) | 2, 90 << 4],
                          [(10 << 4) | 4, 140 << 4, 0]], dtype=torch.int32, device="cuda")
    ncig = torch.tensor([1, 3, 3, 2], dtype=torch.int32, device="cuda")[kind]
    words = table[kind]
    keep = torch.arange(3, device="cuda")[None, :] < ncig[:, None]
    cigar = words[keep].contiguous()
    cigar_off = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    cigar_off[1:] = torch.cumsum(ncig, 0)
    m_per = torch.tensor([150, 148, 150, 140], dtype=torch.int64, device="cuda")[kind]   # M bases per record
    del pick, kind, words, keep

    class D:
        pass
    d = D()
    d.tid, d.pos, d.flag, d.cigar_off, d.cigar = tid, pos, fl, cigar_off, cigar
    d.l_qseq = torch.full((n,), L, dtype=torch.int32, device="cuda")
    d.seq_off = torch.arange(n + 1, device="cuda", dtype=torch.int64) * ((L + 1) // 2)
    d.seq4 = torch.randint(0, 256, (n * ((L + 1) // 2),), device="cuda", generator=g, dtype=torch.uint8)
    n_ops, n_m = int(cigar.numel()), int(((cigar & 15) == 0).sum().item())
    if raw_only:
        _bam_raw_legs(ctx, reps, legs, d, ncig, TL, L, None)
        return
    keep_alive, ts3, ts4, tsw, tsf_ = [], [], [], [], []
    ANY = 0x80000000            # HPN_DEPTH_ANY_ORDER: nothing swept early -> the two kernels of round 2, timed on their own
    for r in range(reps + 1):
        ctx._ck(ctx.L.hpn_depth_begin(ctx.h, 0, TL, 0x704 | ANY), "hpn_depth_begin")
        b = ctx._batch(d, keep_alive)
        ctx._ck(ctx.L.hpn_depth_add_dev(ctx.h, C.byref(b)), "hpn_depth_add_dev")
        ctx.sync()
        t3 = ctx.last_kernel_ms(2)
        runs, win = ctx.depth_finish(TL, 20000, runs_cap=1 << 27)
        t4 = ctx.last_kernel_ms(2)
        if r:
            ts3.append(t3), ts4.append(t4)
    slots = TL + 1 + (1 << 21)
    k3_bytes, k4_bytes = n * 16 + 4 * n_ops + 8 * n_m, slots * 4 + 12 * len(runs) + 8 * len(win)
    legs.append(_leg("K3 k_depth_index + k_depth_tiles: CIGAR-M difference array (bam2depth fetch_func), two-pass route (input in any order)",
                     statistics.median(ts3), k3_bytes, records=n, cigar_ops=n_ops, target_len=TL))
    legs.append(_leg("K4 k_depth_scan: prefix sum + runs + window sums (hash2BedGraph, overlap), two-pass route", statistics.median(ts4),
                     k4_bytes, positions=slots, runs=len(runs)))
    # the default on coordinate-sorted input: tiles the batch has moved beyond are swept by the workgroup that gathered them
    # (k_depth_index + k_depth_sweep in hpn_depth_add, k_depth_scan over what is left in hpn_depth_finish); same results
    for r in range(reps + 1):
        ctx._ck(ctx.L.hpn_depth_begin_w(ctx.h, 0, TL, 0x704, 20000), "hpn_depth_begin_w")
        b = ctx._batch(d, keep_alive)
        ctx._ck(ctx.L.hpn_depth_add_dev(ctx.h, C.byref(b)), "hpn_depth_add_dev")
        ctx.sync()
        t3 = ctx.last_kernel_ms(2)
        runs2, win2 = ctx.depth_finish(TL, 20000, runs_cap=1 << 27)
        t4 = ctx.last_kernel_ms(2)
        if r:
            tsw.append(t3), tsf_.append(t4)
    assert np.array_equal(runs2, runs) and np.array_equal(win2, win)
    del runs2, win2
    legs.append(_leg("K3+K4 k_depth_index + k_depth_sweep (+ k_depth_scan over the last tiles): bam2depth's fetch_func, hash2BedGraph and overlap in one pass "
                     "over sorted records, no difference array through HBM", statistics.median(tsw) + statistics.median(tsf_), k3_bytes + k4_bytes,
                     add_ms=round(statistics.median(tsw), 4), finish_ms=round(statistics.median(tsf_), 4), records=n, runs=len(runs),
                     identical_to_two_pass_route=True))
    tsf = []
    for r in range(reps + 1):
        text_bytes = ctx.depth_bedgraph_format("chr1")
        if r:
            tsf.append(ctx.last_kernel_ms(2))
    legs.append(_leg("k_bedgraph_text: the runs as bedGraph lines, formatted on the device (hash2BedGraph's fprintf)",
                     statistics.median(tsf), 12 * len(runs) + text_bytes, runs=len(runs), text_bytes=int(text_bytes)))
    # size-independent properties of the result: coverage mass = M bases of the kept records, runs sorted and disjoint
    m_bases = int(m_per[(fl & 0x704) == 0].sum().item())
    assert int(win.sum()) == m_bases, (int(win.sum()), m_bases)
    assert int(((runs[:, 1] - runs[:, 0]).astype(np.int64) * runs[:, 2]).sum()) == m_bases
    assert bool((runs[1:, 0] >= runs[:-1, 1]).all()) and bool((runs[:, 2] > 0).all())
    del runs, win
    off = np.array([0, TL // 20000 + 1], np.uint64)
    ts5 = []
    for r in range(reps + 1):
        bins, gc, ln, touched, nc = ctx.window_counts(d, off, 20000, dev=True)
        if r:
            ts5.append(ctx.last_kernel_ms(3))
    counted = int(((fl & 4) == 0).sum().item())
    assert int(bins.sum()) == nc == counted and int(ln.sum()) == counted * L
    # GC of the counted records, nibble by nibble (codes 2 = C and 4 = G), by torch on the same resident bytes
    want_gc, nb = 0, (L + 1) // 2
    for a in range(0, n, 10_000_000):
        b = min(n, a + 10_000_000)
        blk = d.seq4[a * nb:b * nb].view(b - a, nb)
        hi, lo = blk >> 4, blk & 15
        per = ((hi == 2) | (hi == 4)).sum(1) + ((lo == 2) | (lo == 4)).sum(1)
        want_gc += int(per[(fl[a:b] & 4) == 0].sum().item())
        del blk, hi, lo, per
    assert int(gc.sum()) == want_gc, (int(gc.sum()), want_gc)
    legs.append(_leg("K5 k_window_add: per-window count / GC / length (bam_sliding_count fetch_func + cal_GC)",
                     statistics.median(ts5), n * (12 + (L + 1) // 2), bytes_touched=n * (20 + (L + 1) // 2), records=n))   # 8(d): n x (12 B + ceil(l_qseq / 2)); the SoA view also holds flag and the 8-byte seq_off
    try:
        _bam_raw_legs(ctx, reps, legs, d, ncig, TL, L, (bins, gc, ln, nc))
    except Exception as e:  # noqa: BLE001  (the SoA legs above stand on their own)
        legs.append({"kernel": "raw BAM route", "failed": str(e)[:300]})
    del d, tid, pos, fl, cigar, cigar_off, m_per
    torch.cuda.empty_cache()


def _raw_stream(d, ncig, a, b, L):
    """Records a .. b-1 of the SoA batch as BAM records (bam.h:178-187 behind block_size; name "r%09d", mapq 30, mate unset,
    qualities 30) back to back: what BGZF blocks inflate to.  -> (uint8 stream, int64 record sizes)"""
    import torch
    m = b - a
    nb = (L + 1) // 2
    nc_ = ncig[a:b].to(torch.int64)
    size = 36 + 11 + 4 * nc_ + nb + L
    W = 36 + 11 + 12 + nb + L
    mat = torch.zeros((m, W), dtype=torch.uint8, device="cuda")

    def le(col, v, k):
        for j in range(k):
            mat[:, col + j] = ((v >> (8 * j)) & 255).to(torch.uint8)
    le(0, size - 4, 4)
    le(8, d.pos[a:b].to(torch.int64), 4)
    mat[:, 12], mat[:, 13] = 11, 30
    le(16, nc_, 2)
    le(18, d.flag[a:b].to(torch.int64), 2)
    le(20, torch.full((m,), L, dtype=torch.int64, device="cuda"), 4)
    mat[:, 24:32] = 255
    idx = torch.arange(a, b, device="cuda", dtype=torch.int64)
    mat[:, 36] = ord("r")
    for k in range(9):
        mat[:, 37 + k] = (48 + (idx // 10 ** (8 - k)) % 10).to(torch.uint8)
    c0 = d.cigar_off[a:b].to(torch.int64)
    for j in range(3):
        w = d.cigar[torch.clamp(c0 + j, max=d.cigar.numel() - 1)].to(torch.int64)
        w = torch.where(nc_ > j, w, torch.zeros_like(w))
        le(47 + 4 * j, w, 4)
    seq = d.seq4[a * nb:b * nb].view(m, nb)
    for k in (1, 2, 3):
        rows = nc_ == k
        if bool(rows.any()):
            s0 = 47 + 4 * k
            mat[rows, s0:s0 + nb] = seq[rows]
            mat[rows, s0 + nb:s0 + nb + L] = 30
    keep = torch.arange(W, device="cuda")[None, :] < size[:, None]
    return mat[keep], size


def _bam_raw_legs(ctx, reps, legs, d, ncig, TL, L, want_window):
    """The route the BAM tools take: the same chr1-at-30x records held as inflated BAM bytes, indexed in place
    (k_raw_starts / _count / _scan / _index: bam_read1's walk, samtools-0.1.19 bam.c:191) and read in place by the depth and
    the window kernels -- in launches of 4.4 M records (~1.2 GB of inflated bytes, 19 K blocks), as host/bam_gpu.hpp makes them."""
    import torch
    n = int(d.pos.numel())
    per_block, per_launch = 230, 230 * 19_130
    nb = (L + 1) // 2
    parts, sizes = [], []
    for a in range(0, n, 4_000_000):
        st, sz = _raw_stream(d, ncig, a, min(n, a + 4_000_000), L)
        parts.append(st), sizes.append(sz)
    stream = torch.cat(parts + [torch.zeros(4096, dtype=torch.uint8, device="cuda")])
    size = torch.cat(sizes)
    del parts, sizes
    rec_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    rec_off[1:] = torch.cumsum(size, 0)
    stream_len = int(rec_off[-1].item())
    BL = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("out_off", "<u8")])
    launches = []
    for a in range(0, n, per_launch):
        b = min(n, a + per_launch)
        starts = rec_off[a:b:per_block].cpu().numpy()
        ends = np.append(starts[1:], int(rec_off[b].item()))
        tab = np.zeros(len(starts), BL)
        tab["out_off"], tab["out_len"] = starts - starts[0], ends - starts
        launches.append((int(starts[0]), int(ends[-1] - starts[0]), b - a, torch.from_numpy(tab.view(np.uint8).copy()).cuda(),
                         torch.zeros(len(starts), dtype=torch.int32, device="cuda")))
    n_ops = int(d.cigar.numel())
    # ---- the index alone ----
    t_idx = []
    for r in range(reps + 1):
        tot = 0.0
        for off, ln_, cnt, d_tab, d_st in launches:
            info = ctx.bam_raw_index_dev(stream[off:], d_tab, d_tab.numel() // 24, 0, d_st)
            ctx.sync()
            assert info.flags == 0 and info.n_records == cnt, (info.flags, info.n_records, cnt)
            tot += ctx.last_kernel_ms(6)
        if r:
            t_idx.append(tot)
    legs.append(_leg("raw BAM route, record index: k_raw_starts + k_raw_count + k_raw_scan + k_raw_index on inflated BAM bytes (bam_read1's walk)",
                     statistics.median(t_idx), n * 44, bytes_touched=n * (128 + 16), records=n, launches=len(launches), stream_bytes=stream_len,
                     ms_per_launch=round(statistics.median(t_idx) / len(launches), 4),
                     note="algorithmic: 36 B of fixed fields + 8 B of rec_off per record; bytes_touched: a 128-byte line per record + the offset written twice (list, rec_off); "
                          "kernel_ms includes the host's look at the counts between k_raw_scan and k_raw_index (one read-back per launch)"))
    # ---- bam2depth on the raw records: k_depth_index<RawRecs> + k_depth_sweep<RawRecs> per launch, then the finish ----
    t_add, t_fin = [], []
    runs_raw = win_raw = None
    for r in range(reps + 1):
        ctx._ck(ctx.L.hpn_depth_begin_w(ctx.h, 0, TL, 0x704, 20000), "hpn_depth_begin_w")
        tot = 0.0
        for off, ln_, cnt, d_tab, d_st in launches:
            ctx.bam_raw_index_dev(stream[off:], d_tab, d_tab.numel() // 24, 0, d_st)
            ctx._ck(ctx.L.hpn_depth_add_raw_dev(ctx.h, C.c_void_p(stream[off:].data_ptr())), "hpn_depth_add_raw_dev")
            ctx.sync()
            tot += ctx.last_kernel_ms(2)
        runs_raw, win_raw = ctx.depth_finish(TL, 20000, runs_cap=1 << 27)
        if r:
            t_add.append(tot), t_fin.append(ctx.last_kernel_ms(2))
    m_bases = int(win_raw.sum())
    assert int(((runs_raw[:, 1] - runs_raw[:, 0]).astype(np.int64) * runs_raw[:, 2]).sum()) == m_bases
    # the same records through the SoA sweep: the same runs
    same = None
    runs_soa = win_soa = None
    if want_window is not None:
        ctx._ck(ctx.L.hpn_depth_begin_w(ctx.h, 0, TL, 0x704, 20000), "hpn_depth_begin_w")
        keep_alive = []
        bsoa = ctx._batch(d, keep_alive)
        ctx._ck(ctx.L.hpn_depth_add_dev(ctx.h, C.byref(bsoa)), "hpn_depth_add_dev")
        runs_soa, win_soa = ctx.depth_finish(TL, 20000, runs_cap=1 << 27)
        same = bool(np.array_equal(runs_soa, runs_raw) and np.array_equal(win_soa, win_raw))
        assert same, "raw-record route and SoA route give different runs"
    k34 = n * 16 + 4 * n_ops + 8 * int(((d.cigar & 15) == 0).sum().item()) + (TL + 1 + (1 << 21)) * 4 + 12 * len(runs_raw) + 8 * len(win_raw)
    legs.append(_leg("raw BAM route, bam2depth: k_depth_index<RawRecs> + k_depth_sweep<RawRecs> per launch of 4.4 M records (+ k_depth_scan over the last tiles)",
                     statistics.median(t_add) + statistics.median(t_fin), k34, add_ms=round(statistics.median(t_add), 4),
                     finish_ms=round(statistics.median(t_fin), 4), records=n, launches=len(launches), runs=len(runs_raw), identical_to_soa_route=same))
    del runs_raw, win_raw, runs_soa, win_soa
    # ---- bam_sliding_count on the raw records: k_raw_fields + k_window_add (sequence read in place) ----
    off_w = np.array([0, TL // 20000 + 1], np.uint64)
    t_f, t_w = [], []
    res = None
    for r in range(reps + 1):
        ctx._ck(ctx.L.hpn_window_begin(ctx.h, 1, off_w.ctypes.data_as(C.c_void_p), 20000), "hpn_window_begin")
        tf = tw = 0.0
        for off, ln_, cnt, d_tab, d_st in launches:
            ctx.bam_raw_index_dev(stream[off:], d_tab, d_tab.numel() // 24, 0, d_st)
            ctx._ck(ctx.L.hpn_window_add_raw_dev(ctx.h, C.c_void_p(stream[off:].data_ptr())), "hpn_window_add_raw_dev")
            ctx.sync()
            tf += ctx.last_kernel_ms(7)
            tw += ctx.last_kernel_ms(3)
        tot = int(off_w[-1])
        bins, gc, ln = np.zeros(tot, np.uint32), np.zeros(tot, np.uint64), np.zeros(tot, np.uint32)
        touched, ncnt = np.zeros(1, np.uint8), C.c_uint64(0)
        ctx._ck(ctx.L.hpn_window_finish(ctx.h, bins.ctypes.data_as(C.c_void_p), gc.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p),
                                        touched.ctypes.data_as(C.c_void_p), C.byref(ncnt)), "hpn_window_finish")
        res = (bins, gc, ln, ncnt.value)
        if r:
            t_f.append(tf), t_w.append(tw)
    same_w = None
    if want_window is not None:
        wb, wg, wl, wn = want_window
        same_w = bool(np.array_equal(res[0], wb) and np.array_equal(res[1], wg) and np.array_equal(res[2], wl) and res[3] == wn)
        assert same_w, "raw-record route and SoA route give different window counts"
    ms = statistics.median(t_f) + statistics.median(t_w)
    legs.append(_leg("raw BAM route, bam_sliding_count: k_raw_fields + k_window_add with the sequence read in place",
                     ms, n * (12 + nb), bytes_touched=n * (8 + 28 + 24 + 2 * 128), fields_ms=round(statistics.median(t_f), 4),
                     window_add_ms=round(statistics.median(t_w), 4), records=n, launches=len(launches),
                     records_per_ms_window_add=round(n / statistics.median(t_w)), identical_to_soa_route=same_w,
                     note="bytes_touched: rec_off + the field view written and read + the two 128-byte lines a record's fixed part and sequence lie in "
                          "(records of ~280 bytes: 4 of every 5 lines of the stream are fetched whatever is read from them)"))
    del stream, rec_off, size, launches
    torch.cuda.empty_cache()


# --------------------------------------------------------------------------------------------------------------
# exactness of the headline launch
# --------------------------------------------------------------------------------------------------------------
def exact_check(ctx, d_qual, d_off, n, L, seed, first, local_counts):
    """local_counts: what K1 returned for THIS rank's resident batch (before any all-reduce)."""
    import torch
    # (1) an independent kernel over the same resident bytes: K1L builds Quality[128][512]; total / Q20 / Q30 are row sums
    ctx.fastq_tally_dev(d_qual, d_off, n, flags=1)
    full = ctx.fastq_tally_fetch(qual_hist=True)
    qh = np.asarray(full.qual_hist, np.uint64)
    rows = qh.sum(axis=1)
    k1l = (int(rows.sum()), int(rows[53:].sum()), int(rows[63:].sum()))
    k1 = (int(local_counts["total"]), int(local_counts["q20"]), int(local_counts["q30"]))
    assert k1 == k1l, ("K1 and K1L disagree on the resident batch", k1, k1l)
    assert (full.total, full.q20, full.q30) == k1l and np.array_equal(np.asarray(full.seqlen), np.asarray(local_counts["seqlen"]))
    assert int(qh[:, L:].sum()) == 0 and bool((qh[:, :L].sum(axis=0) == n).all())   # every cycle of every read, once
    # (2) three windows of the batch against the CPU oracle (checker use, after the timed region)
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
    orc.orc_counts_new.restype = C.c_void_p
    orc.orc_counts_free.argtypes = [C.c_void_p]
    u8p, u64p = np.ctypeslib.ndpointer(np.uint8, flags="C"), np.ctypeslib.ndpointer(np.uint64, flags="C")
    orc.orc_count_soa.argtypes = [u8p, u64p, C.c_uint64, C.c_void_p]
    orc.orc_counts_flat_quality.argtypes = [C.c_void_p, u64p]
    orc.orc_synth_soa.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u8p, u8p, u64p]
    w = min(200_000, n)
    windows = []
    for a in sorted({0, (n - w) // 2, n - w}):
        q = d_qual[a * L:(a + w) * L].cpu().numpy()
        o = (d_off[a:a + w + 1] - d_off[a]).cpu().numpy().astype(np.uint64)
        box = orc.orc_counts_new()
        assert orc.orc_count_soa(q, o, w, box) == 0
        want = np.zeros(128 * 512, np.uint64)
        orc.orc_counts_flat_quality(box, want)
        orc.orc_counts_free(box)
        want = want.reshape(128, 512)
        ctx.fastq_tally_dev(d_qual[a * L:], d_off[a:a + w + 1] - d_off[a], w, flags=1)
        got = ctx.fastq_tally_fetch(qual_hist=True)
        assert np.array_equal(np.asarray(got.qual_hist, np.uint64), want), f"window at record {a}: Quality matrix differs from the oracle"
        ctx.fastq_tally_dev(d_qual[a * L:], d_off[a:a + w + 1] - d_off[a], w, flags=0)
        fast = ctx.fastq_tally_fetch()
        assert (fast.total, fast.q20, fast.q30) == (int(want.sum()), int(want[53:].sum()), int(want[63:].sum()))
        # and the resident bytes are the generator's (the oracle restates the same counter-based function)
        sq, qq, oo = np.zeros(w * L, np.uint8), np.zeros(w * L, np.uint8), np.zeros(w + 1, np.uint64)
        orc.orc_synth_soa(seed, first + a, w, L, L, sq, qq, oo)
        assert np.array_equal(qq, q)
        windows.append(int(a))
    return {"k1_equals_k1l_on_resident_batch": True, "reads": int(n), "total": k1[0], "q20": k1[1], "q30": k1[2],
            "oracle_windows": {"records_each": int(w), "starts": windows, "matrix_and_counts_identical": True}}


# --------------------------------------------------------------------------------------------------------------
# end to end through the CLI binaries
# --------------------------------------------------------------------------------------------------------------
def _fastq_text(ctx, n, L, seed):
    """n fixed-length records as FASTQ text (names zero-padded: every record has the same byte length)."""
    import torch
    dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(seed, 0, n, L, dq, db, do)
    ctx.sync()
    name_w = 10
    rec = 1 + 1 + name_w + 1 + L + 1 + 2 + L + 1                      # "@r" name \n seq \n +\n qual \n
    t = torch.empty((n, rec), dtype=torch.uint8, device="cuda")
    t[:, 0], t[:, 1] = ord("@"), ord("r")
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    for k in range(name_w):
        t[:, 2 + k] = (48 + (idx // 10 ** (name_w - 1 - k)) % 10).to(torch.uint8)
    p = 2 + name_w
    t[:, p] = 10
    t[:, p + 1:p + 1 + L] = db.view(n, L)
    t[:, p + 1 + L] = 10
    t[:, p + 2 + L], t[:, p + 3 + L] = ord("+"), 10
    t[:, p + 4 + L:p + 4 + 2 * L] = dq.view(n, L)
    t[:, p + 4 + 2 * L] = 10
    return t.cpu().numpy().reshape(-1)


def _gz_members(buf, n_members, threads):
    """Concatenated gzip members (what `cat a.gz b.gz` or bgzip-less pipelines give; gzread reads them as one stream)."""
    cuts = [len(buf) * i // n_members for i in range(n_members + 1)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda i: _gzip_one(buf[cuts[i]:cuts[i + 1]]), range(n_members)))
    return b"".join(parts)


def _gzip_one(b):
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    return co.compress(b) + co.flush()


def _gz_single_member(buf, pieces, threads):
    """ONE gzip member made in parallel the way pigz does: raw deflate pieces ending on sync-flush points, one trailer."""
    cuts = [len(buf) * i // pieces for i in range(pieces + 1)]

    def piece(i):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        b = buf[cuts[i]:cuts[i + 1]]
        return co.compress(b) + co.flush(zlib.Z_FINISH if i == pieces - 1 else zlib.Z_FULL_FLUSH), zlib.crc32(b), len(b)
    with ThreadPoolExecutor(threads) as ex:
        out = list(ex.map(piece, range(pieces)))
    crc = zlib.crc32(buf)
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x04\xff" + b"".join(o[0] for o in out) +
            (crc & 0xffffffff).to_bytes(4, "little") + (len(buf) & 0xffffffff).to_bytes(4, "little"))


def _timed(cmd, cwd, env=None):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1", **(env or {})})
    return time.perf_counter() - t0, p


def _settle():
    """Inputs written a moment ago are still dirty in the page cache, and the kernel's write-back of 16 GB runs beside whatever
    reads them next: a leg timed in those seconds streamed its file at half the rate of the same leg a few seconds later
    (profiles/r05/numa_pagecache_probe.txt: always the second run behind the file's creation).  A file a tool is given has
    normally been on disk for a while: flush first, then time."""
    try:
        os.sync()
    except OSError:
        pass


def _quiet():
    """A GPU process that has just exited is still being taken apart by the driver for 0.1 - 0.2 s, and a process that starts in
    that time pays for it: hipInit 55 -> 120 - 240 ms, the first stream 22 -> 40 - 58 ms (scripts/startup_probe3.sh,
    profiles/r05/startup_pagecache.txt: the first run behind a pause is the fast one, every run right behind another is slow;
    scripts/prof_r05_tools.sh with PAUSE=1 / 0: 0.51 against 0.67 s for the same tool on the same file).  Every timed run of a
    tool here starts a second after the GPU process before it ended: the tool is timed, not its predecessor's teardown."""
    time.sleep(1.0)


def _hpn_lines(p, k=3):
    """The tool's own stage lines (HPN_TIMING=1 on stderr): where the run's wall went, beside the wall itself."""
    return [l[:300] for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("[hpn]")][-k:]


def _outputs(d, inputs):
    return sorted(f for f in os.listdir(d) if f not in inputs and not f.endswith("_hits.png"))


def _exe_dir(td):
    """Where helper programs are built: the scratch directory, unless that is under /dev/shm (mounted noexec)."""
    if os.path.realpath(td).startswith("/dev/shm"):
        d = os.path.join(tempfile.gettempdir(), "hpn_bench_exe")
        os.makedirs(d, exist_ok=True)
        return d
    return td


def host_link(td):
    """What feeds the plain-text legs on THIS box: pinned host -> device copies and pread out of the page cache into pinned memory
    (scripts/micro/h2d_bw.hip, built and run here).  Every end-to-end leg is priced against it: input_GBps / link_GBps = link_frac.
    None when the micro-benchmark cannot be built or run (the legs then carry no link_frac)."""
    exe = os.path.join(_exe_dir(td), "h2d_bw")
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "scripts", "micro", "h2d_bw.hip"), "-o", exe, "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        out = subprocess.run([exe, td], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
        j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        j["h2d_best_GBps"] = max(v for k, v in j.items() if k.startswith("h2d_"))
        j["pread_best_GBps"] = max(v for k, v in j.items() if k.startswith("pread_pagecache"))
        j["pipelined_best_GBps"] = max(v for k, v in j.items() if "pipelined" in k)     # (incl. the pass with the threads next to the device)
        # ... and what ONE output file takes on this box (scripts/micro/write_bw.cpp: slabs of 128 MiB, pwrite on 1 / 4 threads; writes
        # to one file are serialised by the file system, 10 GB/s on the round's boxes): the ceiling of fastq_trim's output
        wexe = os.path.join(_exe_dir(td), "write_bw")
        try:
            subprocess.check_call(["g++", "-O2", os.path.join(ROOT, "scripts", "micro", "write_bw.cpp"), "-o", wexe, "-lpthread"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            for t in (1, 4):
                o = subprocess.run([wexe, os.path.join(td, "write_bw.out"), "4096", str(t), "0"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout.decode()
                j[f"file_write_x{t}_GBps"] = float(o.split(":")[1].split()[0])
            j["file_write_best_GBps"] = max(j["file_write_x1_GBps"], j["file_write_x4_GBps"])
        except Exception:  # noqa: BLE001
            pass
        finally:
            if os.path.exists(wexe):
                os.unlink(wexe)
        return j
    except Exception:  # noqa: BLE001
        return None
    finally:
        if os.path.exists(exe):
            os.unlink(exe)


def _price(leg, in_bytes, link):
    """input_GBps of our run and its share of the box's pipelined pread + H2D rate (a leg cannot beat that from the page cache)."""
    if leg.get("hpngs") and in_bytes:
        leg["input_bytes"] = int(in_bytes)
        leg["input_GBps"] = round(in_bytes / leg["hpngs"]["seconds"] / 1e9, 2)
        # ... and net of the process's fixed cost (HIP init + context + exit, measured on a one-record file: startup_s) -- what the
        # tool sustains while it streams; the runtime's fixed cost is 0.25 - 0.4 s per process on the round's boxes whatever the
        # input (scripts/micro/startup_hip.hip, exit_hip.hip)
        if leg.get("startup_s") and leg["hpngs"]["seconds"] > leg["startup_s"]:
            leg["input_GBps_net_of_startup"] = round(in_bytes / (leg["hpngs"]["seconds"] - leg["startup_s"]) / 1e9, 2)
        if link:
            leg["link_frac"] = round(leg["input_GBps"] / link["pipelined_best_GBps"], 3)
            if leg.get("input_GBps_net_of_startup"):
                leg["link_frac_net_of_startup"] = round(leg["input_GBps_net_of_startup"] / link["pipelined_best_GBps"], 3)
            if leg.get("output_bytes") and link.get("file_write_best_GBps"):      # fastq_trim: the output file is the slower side
                leg["output_GBps"] = round(leg["output_bytes"] / leg["hpngs"]["seconds"] / 1e9, 2)
                leg["file_write_frac"] = round(leg["output_GBps"] / link["file_write_best_GBps"], 3)
    return leg


def _steady_state_legs(ctx, cores, td, pair, L, link=None):
    """Inputs large enough that start-up (HIP init + context + exit, measured on a one-record file and reported as
    startup_s) is a few per cent of the wall: what the tools sustain from the page cache over one PCIe link."""
    import torch
    legs = []
    free = shutil.disk_usage(td).free
    n_small, n_big = 13_000_000, 52_000_000                     # 4.1 GB and 16.3 GB of text
    if free < 60 << 30:
        return [{"leg": "steady state", "skipped": f"{free >> 30} GiB free in {td}"}]
    rec = 14 + 2 * L
    with open(os.path.join(td, "one.fq"), "wb") as f:
        f.write(_fastq_text(ctx, 1, L, 5).tobytes())
    t_start = 1e9
    for _ in range(3):
        _quiet()
        t_start = min(t_start, _timed([os.path.join(BIN, "fastq_count"), "one.fq"], td)[0])
    # 16.3 GB as four generator blocks (host memory stays near 4 GB); the first block alone is small.fq
    with open(os.path.join(td, "big.fq"), "wb") as fb, open(os.path.join(td, "small.fq"), "wb") as fs:
        for k in range(4):
            blk = _fastq_text(ctx, n_small, L, 40 + k)
            fb.write(blk.data)
            if k == 0:
                fs.write(blk.data)
            del blk
    torch.cuda.empty_cache()
    _settle()
    names8 = [f"s{i}.fq" for i in range(8)]
    for nm in names8:
        os.symlink(os.path.join(td, "small.fq"), os.path.join(td, nm))

    def ours_only(label, tool, args, inputs, unit_bases, env=None, compare_to=None, keep_sizes=False):
        wd = tempfile.mkdtemp(prefix="ours_", dir=td)
        for i in inputs:
            os.symlink(os.path.join(td, i), os.path.join(wd, i))
        _timed([os.path.join(BIN, tool)] + args, wd, env)
        for f in _outputs(wd, inputs):
            os.unlink(os.path.join(wd, f))
        _quiet()
        dt, p = _timed([os.path.join(BIN, tool)] + args, wd, env)
        res = {"leg": label, "hpngs": {"seconds": round(dt, 3), "gbases_per_s": round(unit_bases / dt / 1e9, 3), "rc": p.returncode, "tool_lines": _hpn_lines(p)},
               "reference": None, "startup_s": round(t_start, 3), "startup_share": round(t_start / dt, 3)}
        if keep_sizes:     # outputs of many GB: (size, CRC-32) instead of the bytes
            out = {}
            for f in _outputs(wd, inputs):
                crc, size = 0, 0
                with open(os.path.join(wd, f), "rb") as fh:
                    while True:
                        blk = fh.read(64 << 20)
                        if not blk:
                            break
                        crc, size = zlib.crc32(blk, crc), size + len(blk)
                out[f] = (size, crc)
            res["output_bytes"] = sum(v[0] for v in out.values())
        else:
            out = {f: open(os.path.join(wd, f), "rb").read() for f in _outputs(wd, inputs)}
        if compare_to is not None:
            res["outputs_identical_to"] = compare_to[0]
            res["outputs_identical"] = bool(out == compare_to[1])
        shutil.rmtree(wd, ignore_errors=True)
        return res, out

    big_bases, small_bases = n_big * L, n_small * L
    # (fastq_count_kthread: fastq_count prints its rows in completion order, which no two runs share)
    r = pair(f"fastq_count_kthread -t 8, 8 plain files x {n_small:.1e} x {L} bp ({8 * n_small * rec / 1e9:.1f} GB)", "fastq_count_kthread",
             lambda wd: ["-t", "8", "-o", "m.tsv"] + names8, names8, 8 * small_bases)
    r["startup_s"] = round(t_start, 3)
    legs.append(_price(r, 8 * n_small * rec, link))
    r = pair(f"fastq_count, ONE plain file {n_big:.1e} x {L} bp ({n_big * rec / 1e9:.1f} GB; default route: two lanes on the one device)", "fastq_count",
             lambda wd: ["-o", "rep.txt", "big.fq"], ["big.fq"], big_bases)
    r["startup_s"] = round(t_start, 3)
    legs.append(_price(r, n_big * rec, link))
    # the same file by record block over lanes (host/text_shard.hpp).  On this box the lanes share the one device and its one
    # PCIe link; the default above already uses TWO of them for a file of this size (one lane's copy runs beside the other's
    # kernels), HPN_NGPU=1 is the single-context route, HPN_NGPU=4 shows what more lanes on one link cost
    base, base_out = ours_only("fastq_count, the same ONE file on ONE context (HPN_NGPU=1)", "fastq_count",
                               ["-o", "rep.txt", "big.fq"], ["big.fq"], big_bases, env={"HPN_NGPU": "1"})
    lanes, _ = ours_only("fastq_count, the same ONE file by record block over 4 lanes (HPN_NGPU=4, lanes share the device)", "fastq_count",
                         ["-o", "rep.txt", "big.fq"], ["big.fq"], big_bases, env={"HPN_NGPU": "4"}, compare_to=("one context", base_out))
    legs.extend([_price(base, n_big * rec, link), _price(lanes, n_big * rec, link)])
    # ---- fastq_trim at steady state (BASELINE configs[2]'s tool): 16.3 GB in, ~15 GB of trimmed text out to a file ----
    trim_args = ["-i", "big.fq", "-s", "5", "-e", "140", "-o", "t"]
    t1, t1_out = ours_only(f"fastq_trim -s 5 -e 140, ONE plain file {n_big:.1e} x {L} bp ({n_big * rec / 1e9:.1f} GB) -> t.trim.fastq (default route)", "fastq_trim",
                           trim_args, ["big.fq"], big_bases, keep_sizes=True)
    legs.append(_price(t1, n_big * rec, link))
    t4, _ = ours_only("fastq_trim, the same file by record block over 4 lanes (HPN_NGPU=4, lanes share the device), output compared by size + CRC", "fastq_trim",
                      trim_args, ["big.fq"], big_bases, env={"HPN_NGPU": "4"}, compare_to=("the default route", t1_out), keep_sizes=True)
    legs.append(_price(t4, n_big * rec, link))
    os.unlink(os.path.join(td, "big.fq"))
    # the two mates of a paired run (configs[2]: 2 x 150 bp): two fastq_trim processes side by side on the one device
    for nm in ("m1.fq", "m2.fq"):
        os.symlink(os.path.join(td, "small.fq"), os.path.join(td, nm))
    wd = tempfile.mkdtemp(prefix="mates_", dir=td)
    for nm in ("m1.fq", "m2.fq"):
        os.symlink(os.path.join(td, nm), os.path.join(wd, nm))
    _quiet()
    t0 = time.perf_counter()
    ps = [subprocess.Popen([os.path.join(BIN, "fastq_trim"), "-i", nm, "-s", "5", "-e", "140", "-o", nm[:2]], cwd=wd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
          for nm in ("m1.fq", "m2.fq")]
    rcs = [q.wait() for q in ps]
    dt = time.perf_counter() - t0
    same = filecmp.cmp(os.path.join(wd, "m1.trim.fastq"), os.path.join(wd, "m2.trim.fastq"), shallow=False) if rcs == [0, 0] else False
    legs.append(_price({"leg": f"fastq_trim -s 5 -e 140 on the two mates at once: 2 processes x {n_small:.1e} x {L} bp ({2 * n_small * rec / 1e9:.1f} GB in), one device",
                        "hpngs": {"seconds": round(dt, 3), "gbases_per_s": round(2 * small_bases / dt / 1e9, 3), "rc": max(rcs)}, "reference": None,
                        "outputs_identical_to": "each other (the same reads)", "outputs_identical": bool(same), "startup_s": round(t_start, 3)}, 2 * n_small * rec, link))
    shutil.rmtree(wd, ignore_errors=True)
    for nm in ("m1.fq", "m2.fq"):
        os.unlink(os.path.join(td, nm))
    # gzip, one member, 4.1 GB of text x 3 members' worth would take the reference minutes: ours only, checked against the
    # plain-text run of the same reads (which the pair above ties to the reference)
    raw = open(os.path.join(td, "small.fq"), "rb").read()
    with open(os.path.join(td, "gz3.fq.gz"), "wb") as f:
        one = _gz_single_member(raw, 256, cores)
        for _ in range(3):
            f.write(one)
        gz_bytes = 3 * len(one)
    del raw, one
    gz, gz_out = ours_only(f"fastq_count, gzip: 3 members x {n_small:.1e} x {L} bp ({gz_bytes / 1e9:.1f} GB compressed, {3 * n_small * rec / 1e9:.1f} GB of text)",
                           "fastq_count", ["-o", "rep.txt", "gz3.fq.gz"], ["gz3.fq.gz"], 3 * small_bases)
    row = gz_out.get("rep.txt", b"").decode().split("\t")
    gz["counts_closed_form"] = bool(len(row) > 2 and int(row[1]) == 3 * n_small and float(row[2]) == 3 * small_bases)
    gz["text_GBps"] = round(3 * n_small * rec / gz["hpngs"]["seconds"] / 1e9, 2)
    legs.append(_price(gz, gz_bytes, link))
    for nm in names8 + ["small.fq", "gz3.fq.gz", "one.fq"]:
        os.unlink(os.path.join(td, nm))
    return legs


def e2e_legs(ctx, cores, reads=8_000_000, L=150, bam_reads=4_000_000):
    legs = []
    # scratch for the legs' files: the usual temporary directory; where that has less than 62 GiB free (a box whose disk holds
    # somebody's leftovers) and /dev/shm has room, /dev/shm -- said in the first leg ("scratch"), because files there are written
    # without the disk file system's per-file serialisation (the fastq_trim legs' output side)
    base = tempfile.gettempdir()
    try:
        if shutil.disk_usage(base).free < (62 << 30) and os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (120 << 30):
            base = "/dev/shm"
    except OSError:
        pass
    td = tempfile.mkdtemp(prefix="hpn_e2e_", dir=base)
    try:
        link = host_link(td)
        legs.append({"leg": "host link of this box (scripts/micro/h2d_bw.hip): pinned H2D, pread from the page cache, both pipelined", "host_link": link,
                     "scratch": base})
        text = _fastq_text(ctx, reads, L, 31)
        raw = text.tobytes()
        del text
        files = {"plain.fq": raw, "members.fq.gz": _gz_members(raw, 32, cores), "single.fq.gz": _gz_single_member(raw, 64, cores)}
        for name, blob in files.items():
            with open(os.path.join(td, name), "wb") as f:
                f.write(blob)
        sizes = {k: len(v) for k, v in files.items()}
        del files, raw
        _settle()
        bases = reads * L

        def pair(label, tool, args_of, inputs, unit_bases, warm=True):
            """Time our binary and the reference's on the same input in fresh directories; compare every output byte."""
            res, outs = {"leg": label}, {}
            for who, d in (("hpngs", BIN), ("reference", REF)):
                exe = os.path.join(d, tool)
                if not os.access(exe, os.X_OK):
                    res[who] = None
                    continue
                wd = tempfile.mkdtemp(prefix=f"{who}_", dir=td)
                for i in inputs:
                    os.symlink(os.path.join(td, i), os.path.join(wd, i))
                if warm and who == "hpngs":                       # first run pays HIP start-up and the page cache
                    _timed([exe] + args_of(wd), wd)
                    for f in _outputs(wd, inputs):
                        os.unlink(os.path.join(wd, f))
                if who == "hpngs":
                    _quiet()
                dt, p = _timed([exe] + args_of(wd), wd)
                res[who] = {"seconds": round(dt, 3), "gbases_per_s": round(unit_bases / dt / 1e9, 3), "rc": p.returncode}
                if who == "hpngs":
                    res[who]["tool_lines"] = _hpn_lines(p)
                outs[who] = (wd, _outputs(wd, inputs), p.stdout)
            if len(outs) == 2:
                a, b = outs["hpngs"], outs["reference"]
                res["outputs_identical"] = bool(a[1] == b[1] and a[2] == b[2] and all(
                    filecmp.cmp(os.path.join(a[0], f), os.path.join(b[0], f), shallow=False) for f in a[1]))
                res["speedup"] = round(res["reference"]["seconds"] / res["hpngs"]["seconds"], 2)
            for wd, _, _ in outs.values():
                shutil.rmtree(wd, ignore_errors=True)
            return res

        for name, what in (("plain.fq", "plain text"), ("members.fq.gz", "gzip, 32 members"), ("single.fq.gz", "gzip, one member")):
            r = pair(f"fastq_count, {reads:.0e} x {L} bp, {what} ({sizes[name] / 1e6:.0f} MB)", "fastq_count",
                     lambda wd, n=name: ["-o", "rep.txt", n], [name], bases)
            legs.append(r)
        legs.append(pair(f"fastq_count_kthread -L, {reads:.0e} x {L} bp, plain text", "fastq_count_kthread",
                         lambda wd: ["-L", "-o", "m.tsv", "plain.fq"], ["plain.fq"], bases))
        legs.append(pair(f"fastq_trim -s 5 -e 140, {reads:.0e} x {L} bp, plain text -> file", "fastq_trim",
                         lambda wd: ["-i", "plain.fq", "-s", "5", "-e", "140", "-o", "t"], ["plain.fq"], bases))
        for f in ("plain.fq", "members.fq.gz", "single.fq.gz"):
            os.unlink(os.path.join(td, f))
        try:
            legs.extend(_steady_state_legs(ctx, cores, td, pair, L, link))
        except Exception as e:  # noqa: BLE001  (disk or memory of the box: the short legs above stand)
            legs.append({"leg": "steady state", "failed": str(e)[:300]})
        # ---- BAM --------------------------------------------------------------------------------------------------
        synth = os.path.join(_exe_dir(td), "bam_synth")
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "scripts", "bam_synth.cpp"), "-o", synth, "-lz", "-lpthread"])
        contig = bam_reads * 150 // 30 // 2                         # two contigs at 30x
        subprocess.check_call([synth, os.path.join(td, "s.bam"), str(bam_reads), "2", str(contig), str(cores)])
        bsz = os.path.getsize(os.path.join(td, "s.bam"))
        ins = ["s.bam", "s.bam.bai"]
        for tool, args in (("bam2depth", ["-o", "d", "s.bam"]), ("bam2wig", ["-o", "w", "s.bam"]), ("bam_sliding_count", ["-o", "s", "s.bam"])):
            legs.append(pair(f"{tool}, {bam_reads:.0e} x 150 bp over 2 x {contig / 1e6:.0f} Mb (30x), BAM {bsz / 1e6:.0f} MB -> reports", tool,
                             lambda wd, a=args: a, ins, bam_reads * 150))
        try:
            legs.extend(_c4_file_legs(cores, td))
        except Exception as e:  # noqa: BLE001
            legs.append({"leg": "C4 as files", "failed": str(e)[:300]})
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return legs


def _c4_file_legs(cores, td):
    """BASELINE configs[3] as a FILE: the 25 hg38 primary contigs at their true lengths (30x on chr21 + chrM, 3x on the
    rest: 7.0e7 reads, ~10 GB of BAM -- the full 30x file is ~90 GB, more than the box's disk) through the built binaries,
    every output byte against the oracle run on the generator's own records (tests/c4.py; checker use).  The reference
    binaries would need ~3 minutes each on this file: not timed here (their rate is in the 4e6-read legs above)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import c4
    if shutil.disk_usage(td).free < 40 << 30:
        return [{"leg": "C4 as files", "skipped": "less than 40 GiB free"}]
    tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
    n_reads = sum(r for _, _, r in tg)
    t0 = time.perf_counter()
    bam, prefix = c4.synth(td, "hg38.bam", tg, max(2, cores - 1))
    t_synth = time.perf_counter() - t0
    _settle()
    soa = c4.Soa(prefix, len(tg))
    for ext in (".tid", ".pos", ".flag", ".kind", ".seq4"):
        os.unlink(prefix + ext)
    W, legs, want = 20000, [], {}

    def oracle(t):          # (the second bam2depth run reuses the ~1 GB of oracle text)
        if t not in want:
            runs, bins = c4.oracle_depth_target(soa, tg, t, W)
            want[t] = c4.oracle_target_text(tg[t][0], tg[t][1], W, runs, bins) + (len(runs),)
        return want[t]
    shape = f"{n_reads:.2e} x 150 bp over the 25 hg38 contigs (30x chr21 + chrM, 3x the rest), BAM {os.path.getsize(bam) / 1e9:.1f} GB"
    # (defaults on one device: both tools read the file front to back on one worker, four / eight chunks under an inflate launch;
    # HPN_NGPU=3: bam2depth's targets over three workers, bam_sliding_count's record batches to three workers in turn)
    for tool, args, env in (("bam2depth", ["-w", str(W), "-o", "d", "hg38.bam"], {}),
                            ("bam2depth", ["-w", str(W), "-o", "d", "hg38.bam"], {"HPN_NGPU": "3"}),
                            ("bam_sliding_count", ["-w", str(W), "-o", "s", "hg38.bam"], {}),
                            ("bam_sliding_count", ["-w", str(W), "-o", "s", "hg38.bam"], {"HPN_NGPU": "3"})):
        wd = tempfile.mkdtemp(prefix="c4_", dir=td)
        os.symlink(bam, os.path.join(wd, "hg38.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38.bam.bai"))
        _quiet()
        dt, p = _timed([os.path.join(BIN, tool)] + args, wd, env)
        how = {("bam2depth", ""): " (default: one worker)", ("bam2depth", "3"): " with the targets over three workers on the one device (HPN_NGPU=3)",
               ("bam_sliding_count", ""): " (default: one worker)", ("bam_sliding_count", "3"): " with the record batches over three workers on the one device (HPN_NGPU=3)"}[(tool, env.get("HPN_NGPU", ""))]
        leg = {"leg": f"{tool} -w {W}{how}, {shape}",
               "hpngs": {"seconds": round(dt, 3), "gbases_per_s": round(n_reads * 150 / dt / 1e9, 3), "rc": p.returncode,
                         "tool_lines": [l for l in _hpn_lines(p, 40) if "record index" not in l][-3:]}, "reference": None}
        if tool == "bam2depth":
            ok, n_runs = True, 0
            with open(os.path.join(wd, "hg38.bam.1.bedGraph"), "rb") as fb, open(os.path.join(wd, "d.1.depth"), "rb") as fd:
                for t in range(len(tg)):
                    bed, dep, nr = oracle(t)
                    ok = ok and fb.read(len(bed)) == bed and fd.read(len(dep)) == dep
                    n_runs += nr
                ok = ok and fb.read(1) == b"" and fd.read(1) == b""
            leg["outputs_identical_to"], leg["outputs_identical"], leg["bedgraph_lines"] = "oracle (orc_depth_target per target)", bool(ok), n_runs
        else:
            leg["outputs_identical_to"] = "oracle (orc_window_add + float32 replay)"
            leg["outputs_identical"] = bool(open(os.path.join(wd, "s.txt"), "rb").read() == c4.oracle_window_report(soa, tg, W))
        shutil.rmtree(wd, ignore_errors=True)
        legs.append(leg)
    legs[0]["input_made_in_s"] = round(t_synth, 1)
    os.unlink(bam), os.unlink(bam + ".bai")
    return legs
