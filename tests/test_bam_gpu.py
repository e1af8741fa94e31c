"""GPU parity: hpn_depth_* and hpn_window_* (HIP, through the C ABI) vs the oracle's dense
model and vs the reference bam2depth's golden files.  Bit-exact: integers only."""
import numpy as np
import pytest

import orc
from bam_synth import fmt_bedgraph, fmt_depth, make_soa
from conftest import expected, golden_path
from highperformancengs_amd import bamio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _depth_all(ctx, soa, W, mask=0x704):
    bed, dep = b"", b""
    for tid, (name, tlen) in enumerate(soa.refs):
        runs, win = ctx.depth_target(soa, tid, tlen, W, mask)
        assert ctx.depth_bedgraph(name) == fmt_bedgraph(name, runs), name   # the same lines, formatted on the device
        rc, wruns, wbins = orc.depth_target(soa, tid, W, mask)
        assert rc == 0
        assert np.array_equal(runs, wruns), (name, len(runs), len(wruns))
        assert np.array_equal(win.astype(np.float64), wbins), name
        bed += fmt_bedgraph(name, runs)
        dep += fmt_depth(name, tlen, W, win)
    return bed, dep


@pytest.mark.parametrize("case,bam,W,pre", [("depth_a3", "e.bam", 100, "d"), ("depth_rand", "rand.bam", 20000, "r"),
                                            ("depth_rand_w1000", "rand.bam", 1000, "r")])
def test_bam2depth_golden_files(ctx, case, bam, W, pre):
    soa = bamio.read_bam_records(golden_path("bam", bam))
    bed, dep = _depth_all(ctx, soa, W)
    assert bed == expected(case, f"{bam}.1.bedGraph")
    assert dep == expected(case, f"{pre}.1.depth")


@pytest.mark.parametrize("n,seed,W", [(0, 1, 1000), (1, 2, 50), (5000, 3, 20000), (200_000, 4, 20000),
                                      (200_000, 5, 777), (50_000, 6, 1)])
def test_depth_synthetic(ctx, n, seed, W):
    refs = [("chrA", 3_000_000), ("chrB", 1_234_567), ("tiny", 300), ("chrC", 40_000)]
    if W == 1:
        refs = [("chrA", 300_000), ("tiny", 300)]
    soa = make_soa(n, refs, seed)
    _depth_all(ctx, soa, W)
    _depth_all(ctx, soa, W, mask=0x4)  # bam2wig's filter (bam2wig.c:88)


def test_depth_position_zero_and_overhang(ctx):
    refs = [("c", 1000)]
    sam = "@SQ\tSN:c\tLN:1000\n" + "".join(
        f"r{i}\t0\tc\t{p}\t30\t{cg}\t*\t0\t0\t*\t*\n" for i, (p, cg) in enumerate(
            [(1, "10M"), (1, "10M"), (11, "5M"), (11, "5M"), (16, "3M"), (990, "50M"), (995, "5M10D20M"), (1000, "1M")]))
    r, recs, _ = bamio.records_from_sam(sam)
    soa = bamio.BamSoA(refs=r, tid=np.array([x.tid for x in recs], np.int32), pos=np.array([x.pos for x in recs], np.int32),
                       flag=np.zeros(len(recs), np.uint32), l_qseq=np.zeros(len(recs), np.int32),
                       cigar_off=np.concatenate([[0], np.cumsum([len(x.cigar) for x in recs])]).astype(np.uint32),
                       cigar=np.concatenate([x.cigar for x in recs]).astype(np.uint32),
                       seq_off=np.zeros(len(recs) + 1, np.uint64), seq4=np.zeros(1, np.uint8))
    bed, _ = _depth_all(ctx, soa, 100)
    # equal-depth neighbours merge across a breakpoint (bam2depth.c:212-214); runs are not clipped at target_len
    assert bed.startswith(b"c\t0\t15\t2\nc\t15\t18\t1\n")
    assert bed.endswith(b"\t1039\t1\n")  # 990+50M runs to 1039, past LN:1000


def test_depth_incremental_batches_and_capacity(ctx):
    import ctypes as C
    from highperformancengs_amd import _lib
    refs = [("chrA", 2_000_000)]
    soa = make_soa(60_000, refs, 11)
    whole_runs, whole_win = ctx.depth_target(soa, 0, refs[0][1], 5000)
    # same records in three hpn_depth_add calls
    L, keep = ctx.L, []
    assert L.hpn_depth_begin(ctx.h, 0, refs[0][1], 0x704) == 0
    cuts = [0, 20_000, 20_001, 60_000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = bamio.BamSoA(refs=refs, tid=soa.tid[a:b], pos=soa.pos[a:b], flag=soa.flag[a:b], l_qseq=soa.l_qseq[a:b],
                            cigar_off=soa.cigar_off[a:b + 1], cigar=soa.cigar, seq_off=soa.seq_off[a:b + 1], seq4=soa.seq4)
        bb = ctx._batch(part, keep)
        assert L.hpn_depth_add(ctx.h, C.byref(bb)) == 0
    # too small a caller buffer reports the needed size
    small = np.zeros((10, 3), np.int32)
    nr = C.c_uint64(0)
    rc = L.hpn_depth_finish(ctx.h, 5000, small.ctypes.data, 10, C.byref(nr), None)
    assert rc == _lib.E_CAPACITY and nr.value == len(whole_runs)
    runs, win = ctx.depth_finish(refs[0][1], 5000)
    assert np.array_equal(runs, whole_runs) and np.array_equal(win, whole_win)
    # other window size on the same difference array
    runs2, win2 = ctx.depth_finish(refs[0][1], 333)
    assert np.array_equal(runs2, whole_runs) and win2.sum() == whole_win.sum()


FAR_CIGARS = ["150M", "50M2047N50M", "50M2048N50M", "30M1999D20M100N40M", "10M20000N10M30000N10M", "5M100000D5M",
              "1M16383N1M", "40M2I108M", "10S140M", "2049M", "17000M", "1N1M", "40000M"]


@pytest.mark.parametrize("n,seed,sort", [(40_000, 31, True), (40_000, 32, False), (300, 33, True)])
def test_depth_far_breakpoints(ctx, n, seed, sort):
    """Breakpoints far behind a record's pos (long D / N / M: beyond the reach the owner tile gathers) go through the
    counted far path; the result is the dense model's whatever the distances."""
    refs = [("chrA", 1_500_000), ("chrB", 200_000)]
    soa = make_soa(n, refs, seed, sort=sort, cigars=FAR_CIGARS, max_start_frac=0.9)
    _depth_all(ctx, soa, 20000)
    _depth_all(ctx, soa, 313, mask=0x4)


def test_depth_tile_edges(ctx):
    """Records and breakpoints exactly on tile thresholds (16384 k, 16384 k - 2048) and one past / before them."""
    refs = [("c", 120_000)]
    T, R = 16384, 2048
    starts = sorted({max(0, t * T + d) for t in range(0, 7) for d in (-R - 1, -R, -R + 1, -151, -150, -149, -1, 0, 1)} | {0, 1})
    recs = []
    for p in starts:
        for cg in ("150M", "1M", "10M2037N1M", "10M2038N1M", "10M2039N1M", "100M100D100M"):
            recs.append((p, cg))
    soa = bamio.BamSoA(refs=refs, tid=np.zeros(len(recs), np.int32), pos=np.array([p for p, _ in recs], np.int32),
                       flag=np.zeros(len(recs), np.uint32), l_qseq=np.zeros(len(recs), np.int32),
                       cigar_off=np.concatenate([[0], np.cumsum([len(bamio.parse_cigar(c)) for _, c in recs])]).astype(np.uint32),
                       cigar=np.concatenate([bamio.parse_cigar(c) for _, c in recs]).astype(np.uint32),
                       seq_off=np.zeros(len(recs) + 1, np.uint64), seq4=np.zeros(1, np.uint8))
    _depth_all(ctx, soa, 1000)


@pytest.mark.parametrize("W", [1, 7, 1000, 1024, 20000, 70000])
def test_depth_dense_change_points(ctx, W):
    """A change point at (nearly) every position over several sub-tiles of the scan: more runs per 16384 positions than the
    scan stages in LDS (10240), so both its staged and its direct stores run, side by side in one sub-tile; stretches
    without coverage, one run open across a whole sub-tile and more, runs that begin and end on sub-tile and group edges."""
    refs = [("d", 400_000)]
    rng = np.random.default_rng(77)
    pos, cg = [], []
    for p in range(1000, 40_000):                       # one or two 31-base reads per position: coverage changes at EVERY position
        for _ in range(1 + p % 2):
            pos.append(p), cg.append("31M")
    for p in range(40_000, 90_000):                     # a read of random length at every position: ~3 of 4 positions change
        pos.append(p), cg.append("%dM" % rng.integers(1, 40))
    pos.append(95_000), cg.append("60000M")             # one run open across sub-tiles 6 .. 9
    for p in range(100_000, 131_072, 3):                # on top of it: every third position
        pos.append(p), cg.append("2M")
    for e in (16384, 65536, 131072, 196608):            # runs that end / begin exactly on the edges
        for dlt in (-2, -1, 0, 1):
            pos.append(e + dlt + 200_000 - e % 7), cg.append("1M")
        pos.append(e + 150_000 - 100), cg.append("100M")
        pos.append(e + 150_000), cg.append("100M")
    order = np.argsort(np.array(pos), kind="stable")
    pos, cg = np.array(pos, np.int32)[order], [cg[i] for i in order]
    soa = bamio.BamSoA(refs=refs, tid=np.zeros(len(pos), np.int32), pos=pos, flag=np.zeros(len(pos), np.uint32),
                       l_qseq=np.zeros(len(pos), np.int32), cigar_off=np.arange(len(pos) + 1, dtype=np.uint32),
                       cigar=np.array([bamio.parse_cigar(c)[0] for c in cg], np.uint32),
                       seq_off=np.zeros(len(pos) + 1, np.uint64), seq4=np.zeros(1, np.uint8))
    runs, win = ctx.depth_target(soa, 0, refs[0][1], W)
    rc, wruns, wbins = orc.depth_target(soa, 0, W, 0x704)
    assert rc == 0 and len(wruns) > 80_000 and (np.diff(wruns[100:30_000, 0]) == 1).all()
    assert np.array_equal(runs, wruns)
    assert np.array_equal(win.astype(np.float64), wbins)


def test_depth_batches_overlap_and_mix_sorted_with_unsorted(ctx):
    """hpn_depth_add calls whose position ranges overlap (tiles written before are added to), an unsorted call in
    between (global atomics on tiles some of which were never written), far breakpoints in every call."""
    import ctypes as C
    refs = [("chrA", 900_000)]
    parts = [make_soa(30_000, refs, 41, sort=True, cigars=FAR_CIGARS, max_start_frac=0.5),
             make_soa(30_000, refs, 42, sort=False, cigars=FAR_CIGARS, max_start_frac=0.9),
             make_soa(30_000, refs, 43, sort=True, cigars=FAR_CIGARS, max_start_frac=0.9),
             make_soa(5, refs, 44, sort=True, cigars=["150M"], max_start_frac=0.9)]
    L, keep = ctx.L, []
    from highperformancengs_amd import _lib
    # calls in no coordinate order: nothing may be swept early (HPN_DEPTH_ANY_ORDER) ...
    assert L.hpn_depth_begin(ctx.h, 0, refs[0][1], 0x704 | _lib.DEPTH_ANY_ORDER) == 0
    for part in parts:
        bb = ctx._batch(part, keep)
        assert L.hpn_depth_add(ctx.h, C.byref(bb)) == 0
    runs, win = ctx.depth_finish(refs[0][1], 1000)
    # (what happens without the flag once something HAS been swept: tests/test_depth_sweep_gpu.py::test_late_records_are_reported)
    whole = bamio.BamSoA(refs=refs, tid=np.concatenate([p.tid for p in parts]), pos=np.concatenate([p.pos for p in parts]),
                         flag=np.concatenate([p.flag for p in parts]), l_qseq=np.concatenate([p.l_qseq for p in parts]),
                         cigar_off=np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p.cigar_off.astype(np.int64)) for p in parts]))]).astype(np.uint32),
                         cigar=np.concatenate([p.cigar for p in parts]), seq_off=np.zeros(1, np.uint64), seq4=np.zeros(1, np.uint8))
    rc, wruns, wbins = orc.depth_target(whole, 0, 1000, 0x704)
    assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins)
    # a second target on the same context starts from nothing although the array was not cleared
    refs2 = [("x", 1), ("chrB", 400_000)]
    soa2 = make_soa(2_000, refs2, 45)
    runs2, win2 = ctx.depth_target(soa2, 1, refs2[1][1], 1000)
    rc, wruns2, wbins2 = orc.depth_target(soa2, 1, 1000, 0x704)
    assert np.array_equal(runs2, wruns2) and np.array_equal(win2.astype(np.float64), wbins2)


def test_bedgraph_text_names_and_numbers(ctx):
    """Lines of every digit count, names from 1 to 200 characters (beyond 46 the text is not staged in LDS, beyond 64 the
    name travels through memory), no run at all."""
    refs = [("c", 268_000_000)]
    recs = [(0, "1M"), (9, "1M"), (10, "90M"), (99, "901M"), (1000, "9000M"), (99_999, "2M"), (1_000_000, "1M"), (9_999_999, "2M"),
            (100_000_000, "5M"), (268_000_100, "400M")] + [(5_000_000 + k, "70M") for k in range(3000)] + [(6_000_000 + 3 * k, "2M") for k in range(3000)]
    soa = bamio.BamSoA(refs=refs, tid=np.zeros(len(recs), np.int32), pos=np.array([p for p, _ in recs], np.int32),
                       flag=np.zeros(len(recs), np.uint32), l_qseq=np.zeros(len(recs), np.int32),
                       cigar_off=np.arange(len(recs) + 1, dtype=np.uint32),
                       cigar=np.array([bamio.parse_cigar(c)[0] for _, c in recs], np.uint32),
                       seq_off=np.zeros(len(recs) + 1, np.uint64), seq4=np.zeros(1, np.uint8))
    order = np.argsort(soa.pos, kind="stable")
    soa.pos, soa.cigar = soa.pos[order], soa.cigar[order]
    runs, win = ctx.depth_target(soa, 0, refs[0][1], 20000)
    assert len(runs) > 3000 and runs[:, 2].max() == 70      # several tiles of 512 lines, depths of one and two digits
    for name in ("c", "chr1", "x" * 46, "y" * 47, "z" * 64, "w" * 65, "HLA-" + "q" * 196):
        assert ctx.depth_bedgraph(name) == fmt_bedgraph(name, runs), len(name)
    empty = bamio.BamSoA(refs=refs, tid=np.zeros(0, np.int32), pos=np.zeros(0, np.int32), flag=np.zeros(0, np.uint32),
                         l_qseq=np.zeros(0, np.int32), cigar_off=np.zeros(1, np.uint32), cigar=np.zeros(1, np.uint32),
                         seq_off=np.zeros(1, np.uint64), seq4=np.zeros(1, np.uint8))
    runs, win = ctx.depth_target(empty, 0, 1000, 100)
    assert len(runs) == 0 and ctx.depth_bedgraph("c") == b""


def test_depth_domain_error(ctx):
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    refs = [("big", 300_000_000)]
    soa = make_soa(10, refs, 1, cigars=["100M"])
    soa.pos[:] = np.arange(10) * 1000 + 268_435_400  # ends beyond 2^28: the reference's 28-bit keys alias
    with pytest.raises(hp.HpnError) as e:
        ctx.depth_target(soa, 0, refs[0][1], 20000)
    assert e.value.status == _lib.E_DOMAIN


# ---- bam_sliding_count ------------------------------------------------------------------

def _window_check(ctx, soa, W):
    rc, off, wb, wg, wl, wt, wn = orc.window_counts(soa, W)
    assert rc == 0
    bins, gc, ln, touched, nc = ctx.window_counts(soa, off, W)
    assert np.array_equal(bins, wb) and np.array_equal(gc, wg) and np.array_equal(ln, wl)
    assert np.array_equal(touched, wt) and nc == wn
    return off, bins, gc, ln, touched


def test_window_appendix_a3(ctx):
    soa = bamio.read_bam_records(golden_path("bam", "e.bam"))
    off, bins, gc, ln, touched = _window_check(ctx, soa, 100)
    # c1 has 7 reads / 60 bases, window 1 holds 5 reads; c2 3 reads / 30 bases
    assert bins[:11].sum() == 7 and ln[:11].sum() == 60 and bins[0] == 5 and bins[1] == 1
    assert bins[11:].sum() == 3 and ln[11:].sum() == 30
    assert "%f" % (np.float32(gc[0]) / np.float32(ln[0]) * np.float32(100)) == "48.888889"


@pytest.mark.parametrize("n,seed,W,sort", [(0, 1, 1000, True), (1, 2, 50, True), (5000, 3, 20000, True),
                                           (300_000, 4, 20000, True), (300_000, 5, 500, True),
                                           (100_000, 6, 1000, False), (77_777, 7, 64, False)])
def test_window_synthetic(ctx, n, seed, W, sort):
    refs = [("chrA", 3_000_000), ("chrB", 1_234_567), ("tiny", 300), ("chrC", 40_000)]
    soa = make_soa(n, refs, seed, sort=sort)
    _window_check(ctx, soa, W)


@pytest.mark.parametrize("cigar", ["150M", "151M", "36M", "75M", "250M", "300M", "16M", "1M", "31M", "33M"])
@pytest.mark.parametrize("sort", [True, False])
def test_window_reads_of_one_length(ctx, cigar, sort):
    """Every read of the batch has the same length: the kernel's lanes then share the records of a step by pieces (P lanes
    per record) and the next pass's pieces are loaded into LDS while this one is counted -- GC sums against the oracle for
    piece counts 1 .. 8 and beyond (300 bases: the generic path), odd lengths (padding nibble), whole spans of 1024 records
    and a ragged last one."""
    refs = [("chrA", 2_000_000), ("chrB", 700_000)]
    soa = make_soa(20_011, refs, 17, sort=sort, cigars=[cigar])
    _window_check(ctx, soa, 1000)


@pytest.mark.parametrize("cigar,W", [("150M", 20000), ("151M", 1000), ("36M", 50), ("250M", 7)])
def test_window_every_nibble_code_and_window_seams(ctx, cigar, W):
    """The GC test is a bit formula on packed nibbles (C = 2, G = 4 of =ACMGRSVTWYHKDBN, bam.h:260): all sixteen codes in every
    position, on the pass the kernel is built around (one target, one window, one length per 64 records) and on its seams -- small
    windows put several windows under one pass, which the second kernel takes (bam_sliding_count.c:106-121)."""
    refs = [("chrA", 1_500_000), ("chrB", 400_000)]
    soa = make_soa(30_011, refs, 23, sort=True, cigars=[cigar])
    rng = np.random.default_rng(5)
    soa.seq4[:] = rng.integers(0, 256, soa.seq4.size, dtype=np.uint8)
    _window_check(ctx, soa, W)
    # flags that skip records (4) in runs and singly, and a stretch of unmapped records (tid -1) in the middle
    soa.flag[1000:1200] = 4
    soa.flag[5000:9000:7] |= 4
    soa.tid[20_000:20_100] = -1
    _window_check(ctx, soa, W)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 70, 109, 110, 128, 1024 + 68, 5 * 1024 + 100, 7 * 1024 + 1023])
def test_window_every_tail_of_a_span(ctx, n):
    """A wave takes 1,024 records in passes of 64; the last span of a batch is ragged.  A pass's record count must never exceed 64
    (round 4 counted the GC of the records behind a pass twice when 65..109 records were left: found by the hg38-shaped file test
    on the host route, whose batch ended 68 records into a span)."""
    refs = [("chrA", 900_000)]
    soa = make_soa(n, refs, 41 + n, sort=True, cigars=["150M"])
    _window_check(ctx, soa, 20000)
    _window_check(ctx, soa, 1000)


def test_window_index_wraps_like_unsigned_short(ctx):
    # target_len / W + 1 > 65536: (unsigned short)(pos / W) wraps (bam_sliding_count.c:117)
    refs = [("long", 10_000_000)]
    soa = make_soa(20_000, refs, 9)
    off, bins, *_ = _window_check(ctx, soa, 100)
    assert int(off[-1]) == 100_001 and bins[65536:].sum() == 0 and bins.sum() > 0


def test_window_domain_error(ctx):
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    refs = [("c", 1000)]
    soa = make_soa(10, refs, 1)
    soa.pos[3] = 5000  # beyond the contig: window 50 of 11 -> the reference writes out of bounds
    with pytest.raises(hp.HpnError) as e:
        ctx.window_counts(soa, orc.window_offsets(refs, 100), 100)
    assert e.value.status == _lib.E_DOMAIN


def test_device_resident_batch(ctx):
    """Device-resident SoA (the _dev entry points) on 2e6 records vs the oracle."""
    import torch
    refs = [("chr1", 30_000_000), ("chr2", 20_000_000)]
    soa = make_soa(2_000_000, refs, 21)

    class Dev:
        pass
    d = Dev()
    for f in ("tid", "pos", "l_qseq"):
        setattr(d, f, torch.from_numpy(getattr(soa, f)).cuda())
    d.flag = torch.from_numpy(soa.flag.view(np.int32)).cuda()
    d.cigar_off = torch.from_numpy(soa.cigar_off.view(np.int32)).cuda()
    d.cigar = torch.from_numpy(soa.cigar.view(np.int32)).cuda()
    d.seq_off = torch.from_numpy(soa.seq_off.view(np.int64)).cuda()
    d.seq4 = torch.from_numpy(soa.seq4).cuda()
    for tid, (name, tlen) in enumerate(refs):
        runs, win = ctx.depth_target(d, tid, tlen, 20000, dev=True)
        rc, wruns, wbins = orc.depth_target(soa, tid, 20000)
        assert np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins)
    rc, off, wb, wg, wl, wt, wn = orc.window_counts(soa, 20000)
    bins, gc, ln, touched, nc = ctx.window_counts(d, off, 20000, dev=True)
    assert np.array_equal(bins, wb) and np.array_equal(gc, wg) and np.array_equal(ln, wl) and nc == wn
