"""GPU: the BASELINE.json configurations at their full single-GPU sizes, held to the oracle / to exact properties.

  C2  fastq_count, 1e9 x 150 bp resident (158 GB): K1's counts = K1L's row sums on the same bytes, three 2e5-read windows
      of the batch = the CPU oracle (Quality matrix and counts), the bytes = the generator's.
  C3  fastq_trim, one mate of 5e8 x 150 bp halved to what fits beside its output (2.5e8): output offsets in closed form,
      every output byte against the strided view of the input ON THE DEVICE, oracle on windows.
  C4  bam2depth + bam_sliding_count on a chr1-sized target (248,956,422 bp) at 30x = 4.98e7 records: runs and window sums
      against the oracle's dense model of the whole target (a 1 GB difference array on the CPU), per-window count / GC /
      length against the oracle; plus the domain edge: a breakpoint at 2^28 - 1 is kept, one at 2^28 is refused.
The 8-GPU configuration (C5) needs hardware this box does not have; its arithmetic is covered by tests/test_shard_gloo.py."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def test_c2_fastq_count_1e9_reads_exact(ctx):
    import torch
    import bench_extra
    free, _ = torch.cuda.mem_get_info()
    n, L = 1_000_000_000, 150
    if free < n * (L + 8) + (4 << 30):
        pytest.skip("not enough free HBM for the 158 GB resident batch")
    d_qual = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(12345, 0, n, L, d_qual, None, d_off)
    ctx.fastq_tally_dev(d_qual, d_off, n, flags=0)
    k1 = ctx.fastq_tally_fetch()
    assert k1.total == n * L and int(k1.seqlen[L]) == n and int(k1.seqlen.sum()) == n
    local = {"seqlen": k1.seqlen.copy(), "total": k1.total, "q20": k1.q20, "q30": k1.q30}
    res = bench_extra.exact_check(ctx, d_qual, d_off, n, L, 12345, 0, local)
    assert res["k1_equals_k1l_on_resident_batch"] and res["oracle_windows"]["matrix_and_counts_identical"]
    assert len(res["oracle_windows"]["starts"]) == 3
    del d_qual, d_off
    torch.cuda.empty_cache()


def test_c3_fastq_trim_one_mate_full_size(ctx):
    """BASELINE configs[2]: 5e8 reads of 150 bp per mate.  In + out of a whole mate (150 + 4 + 135 GB) do not fit the 288 GB of
    one device together, so the mate goes through in two slabs of 2.5e8 reads (the second slab's generator continues where the
    first one's ended; the first slab's output is checked and freed before the second is made): offsets in closed form, every
    output byte against the strided view of its input on the device, the oracle on windows at both ends of each slab, across
    the 2^32-byte mark, and ACROSS THE SEAM (the last reads of slab 0 and the first of slab 1 against the oracle run on the CPU
    generator's records for that stretch)."""
    import torch
    n_mate, L, S, E, seed = 500_000_000, 150, 5, 140, 4242
    n = n_mate // 2
    w = 100_000
    seam = {}
    for slab_no in (0, 1):
        first = slab_no * n
        dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
        db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
        do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        ctx.synth_fastq_dev(seed, first, n, L, dq, db, do)
        oq = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
        ob = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
        oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        ctx.fastq_trim_dev(db, dq, do, n, S, E, ob, oq, oo)
        ctx.sync()
        # offsets: closed form (beyond 2^32 bytes: 3.4e10)
        assert int(oo[0].item()) == 0 and int(oo[-1].item()) == n * (E - S) > 1 << 34
        step = torch.arange(0, n + 1, 1 << 16, device="cuda", dtype=torch.int64)
        assert torch.equal(oo[step], step * (E - S))
        assert bool((oo[1:] - oo[:-1] == E - S).all())
        # every output byte = the strided view of the input, compared on the device in pieces
        piece = 10_000_000
        for a in range(0, n, piece):
            b = min(n, a + piece)
            assert torch.equal(oq[a * (E - S):b * (E - S)].view(b - a, E - S), dq[a * L:b * L].view(b - a, L)[:, S:E])
            assert torch.equal(ob[a * (E - S):b * (E - S)].view(b - a, E - S), db[a * L:b * L].view(b - a, L)[:, S:E])
        # oracle on windows (first, one straddling the 2^32-byte mark of the input, last)
        for a in (0, (1 << 32) // L - w // 2, n - w):
            seq = db[a * L:(a + w) * L].cpu().numpy()
            qual = dq[a * L:(a + w) * L].cpu().numpy()
            off = (do[a:a + w + 1] - do[a]).cpu().numpy().astype(np.uint64)
            rc, wseq, wqual, woff = orc.trim_soa(seq, qual, off, S, E)
            assert rc == 0
            assert np.array_equal(oq[a * (E - S):(a + w) * (E - S)].cpu().numpy(), wqual)
            assert np.array_equal(ob[a * (E - S):(a + w) * (E - S)].cpu().numpy(), wseq)
            assert np.array_equal((oo[a:a + w + 1] - oo[a]).cpu().numpy().astype(np.uint64), woff)
        # what lies at the seam: the last w / 2 reads of slab 0, the first w / 2 of slab 1
        h = w // 2
        lo, hi = (n - h, n) if slab_no == 0 else (0, h)
        seam[slab_no] = (oq[lo * (E - S):hi * (E - S)].cpu().numpy(), ob[lo * (E - S):hi * (E - S)].cpu().numpy())
        del dq, db, do, oq, ob, oo
        torch.cuda.empty_cache()
    # the seam window from the CPU generator (records n - w/2 .. n + w/2 of the mate) through the oracle
    seq, qual, off = orc.synth_soa(seed, n - w // 2, w, L, L)
    rc, wseq, wqual, woff = orc.trim_soa(seq, qual, off, S, E)
    assert rc == 0 and int(woff[-1]) == w * (E - S)
    assert np.array_equal(np.concatenate([seam[0][0], seam[1][0]]), wqual)
    assert np.array_equal(np.concatenate([seam[0][1], seam[1][1]]), wseq)


def _chr1_soa():
    """4.98e7 records over a chr1-sized target, SURVEY §8d CIGAR / flag mix, coordinate-sorted."""
    from highperformancengs_amd.bamio import BamSoA, parse_cigar
    TL, L = 248_956_422, 150
    n = 30 * TL // L
    rng = np.random.default_rng(2024)
    pos = np.sort(rng.integers(0, TL - 200, n, dtype=np.int64)).astype(np.int32)
    flags = np.array([0, 16] * 9 + [4, 256, 512, 1024], np.uint32)[rng.integers(0, 22, n)]
    pick = rng.integers(0, 20, n)
    kind = np.where(pick < 17, 0, pick - 16)
    sets = [np.array(parse_cigar(c), np.uint32) for c in ("150M", "40M2I108M", "60M5D90M", "10S140M")]
    ncig = np.array([len(s) for s in sets], np.int64)[kind]
    cigar_off = np.zeros(n + 1, np.uint32)
    np.cumsum(ncig, out=cigar_off[1:])
    cigar = np.zeros(int(cigar_off[-1]), np.uint32)
    for k, s in enumerate(sets):
        idx = np.nonzero(kind == k)[0]
        base = cigar_off[idx].astype(np.int64)
        for j, w in enumerate(s):
            cigar[base + j] = w
    nb = (L + 1) // 2
    seq_off = (np.arange(n + 1, dtype=np.uint64) * nb)
    seq4 = rng.integers(0, 256, n * nb, dtype=np.uint8)
    return BamSoA(refs=[("chr1", TL)], tid=np.zeros(n, np.int32), pos=pos, flag=flags, l_qseq=np.full(n, L, np.int32),
                  cigar_off=cigar_off, cigar=cigar, seq_off=seq_off, seq4=seq4)


def test_c4_chr1_sized_target_against_the_dense_model(ctx):
    soa = _chr1_soa()
    TL, W = soa.refs[0][1], 20000
    rc, wruns, wbins = orc.depth_target(soa, 0, W, 0x704)      # 1 GB difference array on the CPU
    assert rc == 0 and len(wruns) > 50_000_000
    runs, win = ctx.depth_target(soa, 0, TL, W, 0x704, runs_cap=len(wruns) + 16)
    assert runs.shape == wruns.shape and np.array_equal(runs, wruns)
    assert np.array_equal(win.astype(np.float64), wbins)
    # the same records in 25 calls (the tools' batches): tiles written by one call are added to by the next
    import ctypes as C
    from highperformancengs_amd import bamio
    L_, keep = ctx.L, []
    assert L_.hpn_depth_begin(ctx.h, 0, TL, 0x704) == 0
    cuts = np.linspace(0, len(soa.tid), 26).astype(np.int64)
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = bamio.BamSoA(refs=soa.refs, tid=soa.tid[a:b], pos=soa.pos[a:b], flag=soa.flag[a:b], l_qseq=soa.l_qseq[a:b],
                            cigar_off=soa.cigar_off[a:b + 1], cigar=soa.cigar, seq_off=soa.seq_off[a:b + 1], seq4=soa.seq4)
        bb = ctx._batch(part, keep)
        assert L_.hpn_depth_add(ctx.h, C.byref(bb)) == 0
        keep.clear()
    runs2, win2 = ctx.depth_finish(TL, W, runs_cap=len(wruns) + 16)
    assert np.array_equal(runs2, wruns) and np.array_equal(win2, win)
    del runs, runs2, wruns
    # bam_sliding_count on the same records
    rc, off, wb, wg, wl, wt, wn = orc.window_counts(soa, W)
    assert rc == 0
    bins, gc, ln, touched, nc = ctx.window_counts(soa, off, W)
    assert np.array_equal(bins, wb) and np.array_equal(gc, wg) and np.array_equal(ln, wl) and nc == wn


def test_c4_position_key_limit(ctx):
    """The reference's keys keep 28 bits of a position (hashtbl.c:243-249): a breakpoint at 2^28 - 1 is the last one it can hold."""
    import highperformancengs_amd as hp
    from bam_synth import make_soa
    from highperformancengs_amd import _lib
    refs = [("big", 300_000_000)]
    soa = make_soa(4, refs, 1, cigars=["100M"])
    soa.flag[:] = 0
    soa.pos[:] = [5, (1 << 28) - 200, (1 << 28) - 101, (1 << 28) - 101]     # last two end at 2^28 - 1
    runs, win = ctx.depth_target(soa, 0, refs[0][1], 20000)
    assert runs[-1].tolist() == [(1 << 28) - 101, (1 << 28) - 1, 2] or runs[-1][1] == (1 << 28) - 1
    rc, wruns, wbins = orc.depth_target(soa, 0, 20000, 0x704)
    assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins)
    soa.pos[3] = (1 << 28) - 100                                            # ends at 2^28: aliases in the reference
    with pytest.raises(hp.HpnError) as e:
        ctx.depth_target(soa, 0, refs[0][1], 20000)
    assert e.value.status == _lib.E_DOMAIN
    assert orc.depth_target(soa, 0, 20000, 0x704)[0] != 0
