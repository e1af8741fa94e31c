"""GPU parity: bam2depth's record loop swept WHILE the records come (k_depth_sweep: the tile a batch's records have moved
beyond is prefix-summed and turned into runs / window sums by the workgroup that gathered its breakpoints, the look-back
chain carried across hpn_depth_add calls) against the oracle's dense model (bam2depth.c:86-110, 203-236, 132-176).
Whatever the batching, the window size told or not, batches that cannot be swept (far breakpoints, unsorted inside) in
between: the runs and window sums are the oracle's, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import orc
from bam_synth import make_soa
from highperformancengs_amd import _lib, bamio

pytestmark = pytest.mark.gpu
FAR = ["150M", "50M2047N50M", "50M2048N50M", "30M1999D20M100N40M", "10M20000N10M30000N10M", "5M100000D5M"]
# (SoA batches are taken by the sweep when (most operations of a record) x (longest M / D / N operation) <= 2048: NEAR keeps
# that bound, EDGE breaks it although no breakpoint is far -- such batches must simply go the two-pass way)
NEAR = ["150M", "40M2I108M", "60M5D90M", "10S140M", "1M", "70M500D70M", "5=5X", "20M3000I20M", "600M"]
EDGE = ["150M", "40M2I108M", "50M1900N50M", "70M500D70M", "5=5X"]


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _part(soa, a, b):
    return bamio.BamSoA(refs=soa.refs, tid=soa.tid[a:b], pos=soa.pos[a:b], flag=soa.flag[a:b], l_qseq=soa.l_qseq[a:b],
                        cigar_off=soa.cigar_off[a:b + 1], cigar=soa.cigar, seq_off=soa.seq_off[a:b + 1], seq4=soa.seq4)


def _feed(ctx, soa, tid, cuts, W_begin, mask=0x704, dev=False):
    keep = []
    tlen = soa.refs[tid][1]
    rc = ctx.L.hpn_depth_begin_w(ctx.h, tid, tlen, mask, W_begin) if W_begin is not None else ctx.L.hpn_depth_begin(ctx.h, tid, tlen, mask)
    assert rc == 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        bb = ctx._batch(_part(soa, a, b), keep)
        assert ctx.L.hpn_depth_add(ctx.h, C.byref(bb)) == 0, ctx.L.hpn_ctx_last_error(ctx.h)
    return keep


def _swept(ctx):
    v = C.c_uint64(0)
    assert ctx.L.hpn_depth_progress(ctx.h, C.byref(v)) == 0
    return v.value


def _check(ctx, soa, tid, W, mask=0x704):
    runs, win = ctx.depth_finish(soa.refs[tid][1], W)
    rc, wruns, wbins = orc.depth_target(soa, tid, W, mask)
    assert rc == 0
    assert len(runs) == len(wruns) and np.array_equal(runs, wruns)
    assert np.array_equal(win.astype(np.float64), wbins)
    return runs


@pytest.mark.parametrize("n,seed,W,pieces", [(300_000, 1, 20000, 1), (300_000, 2, 20000, 7), (300_000, 3, 1000, 40), (50_000, 4, 37, 13),
                                             (2_000, 5, 20000, 100), (300_000, 6, 5000, 3)])
def test_sorted_stream_any_batching(ctx, n, seed, W, pieces):
    refs = [("chrA", 5_000_000), ("chrB", 3_000_000 + 12_345)]
    soa = make_soa(n, refs, seed, cigars=NEAR)
    rng = np.random.default_rng(seed)
    for tid in (0, 1):
        cuts = [0] + sorted(int(x) for x in rng.integers(0, n, pieces - 1)) + [n]      # (cuts anywhere: other targets' records ride along)
        _feed(ctx, soa, tid, cuts, W)
        # the sweep really took place: everything in front of the tile the last record of the target lies in is final already
        mine = soa.pos[soa.tid == tid]
        assert _swept(ctx) == (min(int(mine.max()), refs[tid][1] + (1 << 21) - 16384) // 16384) * 16384, (tid, pieces)
        runs = _check(ctx, soa, tid, W)
        assert len(runs) > 0
        # another window size on the same result: sums from the runs
        _check(ctx, soa, tid, 777)
        _check(ctx, soa, tid, W)
        # window size not told at all
        _feed(ctx, soa, tid, cuts, None)
        _check(ctx, soa, tid, W)
        # bam2wig's filter
        _feed(ctx, soa, tid, cuts, W, mask=0x4)
        _check(ctx, soa, tid, W, mask=0x4)


def test_a_batch_beyond_the_reach_bound_goes_the_two_pass_way(ctx):
    refs = [("chrA", 2_000_000)]
    soa = make_soa(100_000, refs, 51, cigars=EDGE)
    _feed(ctx, soa, 0, [0, 30_000, 100_000], 20000)
    assert _swept(ctx) == 0
    _check(ctx, soa, 0, 20000)
    # ... and with HPN_DEPTH_ANY_ORDER nothing is swept whatever the batch
    soa = make_soa(100_000, refs, 52, cigars=NEAR)
    _feed(ctx, soa, 0, [0, 30_000, 100_000], 20000, mask=0x704 | _lib.DEPTH_ANY_ORDER)
    assert _swept(ctx) == 0
    _check(ctx, soa, 0, 20000)


def test_batches_the_sweep_must_leave_alone(ctx):
    """Far breakpoints (beyond 2048 of their record's pos) and batches unsorted inside go the two-pass way, between swept
    batches, as long as they do not reach behind the frontier."""
    refs = [("chrA", 6_000_000)]
    third = 2_000_000
    a = make_soa(60_000, refs, 21, cigars=NEAR, max_start_frac=1 / 3)                  # swept
    b = make_soa(40_000, refs, 22, cigars=FAR, max_start_frac=1 / 3)                   # far breakpoints, from `third` on
    b.pos += third
    c = make_soa(40_000, refs, 23, cigars=NEAR, sort=False, max_start_frac=0.1)        # unsorted inside, further on
    c.pos += 2 * third - 300_000
    d = make_soa(60_000, refs, 24, cigars=NEAR, max_start_frac=0.2)                    # swept again; closes everything before
    d.pos += 2 * third
    parts = [a, b, c, d]
    whole = bamio.BamSoA(refs=refs, tid=np.concatenate([p.tid for p in parts]), pos=np.concatenate([p.pos for p in parts]),
                         flag=np.concatenate([p.flag for p in parts]), l_qseq=np.concatenate([p.l_qseq for p in parts]),
                         cigar_off=np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p.cigar_off.astype(np.int64)) for p in parts]))]).astype(np.uint32),
                         cigar=np.concatenate([p.cigar for p in parts]), seq_off=np.zeros(1, np.uint64), seq4=np.zeros(1, np.uint8))
    keep = []
    for W_begin in (5000, None):
        rc = ctx.L.hpn_depth_begin_w(ctx.h, 0, refs[0][1], 0x704, W_begin or 0)
        assert rc == 0
        for p in parts:
            bb = ctx._batch(p, keep)
            assert ctx.L.hpn_depth_add(ctx.h, C.byref(bb)) == 0
        runs, win = ctx.depth_finish(refs[0][1], 5000)
        rc, wruns, wbins = orc.depth_target(whole, 0, 5000, 0x704)
        assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins)


def test_late_records_are_reported(ctx):
    refs = [("chrA", 4_000_000)]
    a = make_soa(50_000, refs, 31, cigars=NEAR)
    keep = []
    assert ctx.L.hpn_depth_begin_w(ctx.h, 0, refs[0][1], 0x704, 1000) == 0
    for part in (_part(a, 25_000, 50_000), _part(a, 0, 25_000)):      # second half first
        bb = ctx._batch(part, keep)
        assert ctx.L.hpn_depth_add(ctx.h, C.byref(bb)) == 0
    nr = C.c_uint64(0)
    assert ctx.L.hpn_depth_finish(ctx.h, 1000, None, 0, C.byref(nr), None) == _lib.E_STATE
    assert b"HPN_DEPTH_ANY_ORDER" in ctx.L.hpn_ctx_last_error(ctx.h)
    # the same calls with the flag: the oracle's result
    assert ctx.L.hpn_depth_begin_w(ctx.h, 0, refs[0][1], 0x704 | _lib.DEPTH_ANY_ORDER, 1000) == 0
    for part in (_part(a, 25_000, 50_000), _part(a, 0, 25_000)):
        bb = ctx._batch(part, keep)
        assert ctx.L.hpn_depth_add(ctx.h, C.byref(bb)) == 0
    _check(ctx, a, 0, 1000)


def test_finish_add_finish(ctx):
    """hpn_depth_finish between batches reports the target so far; more records may follow."""
    refs = [("chrA", 3_000_000)]
    soa = make_soa(90_000, refs, 41, cigars=NEAR)
    keep = _feed(ctx, soa, 0, [0, 30_000, 60_000], 20000)
    first = bamio.BamSoA(refs=refs, tid=soa.tid[:60_000], pos=soa.pos[:60_000], flag=soa.flag[:60_000], l_qseq=soa.l_qseq[:60_000],
                         cigar_off=soa.cigar_off[:60_001], cigar=soa.cigar, seq_off=soa.seq_off[:60_001], seq4=soa.seq4)
    _check(ctx, first, 0, 20000)
    bb = ctx._batch(_part(soa, 60_000, 90_000), keep)
    assert ctx.L.hpn_depth_add(ctx.h, C.byref(bb)) == 0
    _check(ctx, soa, 0, 20000)
    _check(ctx, soa, 0, 20000)


def test_dense_and_empty_stretches(ctx):
    """Every position a change point (the runs' staging area overflows into direct stores), then megabases without a record
    (tiles swept with nothing in them), then records again; coverage carried across the gap."""
    refs = [("c", 9_000_000)]
    n1 = 120_000
    pos = np.concatenate([np.arange(n1, dtype=np.int32) + 1000, np.array([200_000], np.int32), np.arange(50_000, dtype=np.int32) * 3 + 8_000_000])
    cig = np.concatenate([np.full(n1, (2 << 4) | 0, np.uint32), np.array([(7_900_000 << 4) | 0], np.uint32), np.full(50_000, (5 << 4) | 0, np.uint32)])
    n = len(pos)
    soa = bamio.BamSoA(refs=refs, tid=np.zeros(n, np.int32), pos=pos, flag=np.zeros(n, np.uint32), l_qseq=np.zeros(n, np.int32),
                       cigar_off=np.arange(n + 1, dtype=np.uint32), cigar=cig, seq_off=np.zeros(n + 1, np.uint64), seq4=np.zeros(1, np.uint8))
    # (the one long M block is a far breakpoint: its batch goes the two-pass way; the batches around it are swept)
    for cuts in ([0, n], [0, 40_000, n1, n1 + 1, n1 + 20_000, n]):
        _feed(ctx, soa, 0, cuts, 20000)
        _check(ctx, soa, 0, 20000)
