"""GPU parity: hpn_fastq_tally (HIP, through the C ABI) vs the oracle's count_read restatement.

Bit-exact: all outputs are integer counters.
"""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _check(ctx, qual, off, base=None, full=True):
    rc, want = orc.count_soa(qual, off)
    assert rc == 0
    s = want.summary()
    fast = ctx.fastq_tally(qual, off)
    assert np.array_equal(fast.seqlen, want.seqlen)
    assert (fast.total, fast.q20, fast.q30) == (s.sum, s.q20, s.q30)
    if full:
        got = ctx.fastq_tally(qual, off, base=base, qual_hist=True, nuc_hist=base is not None)
        assert np.array_equal(got.seqlen, want.seqlen)
        assert (got.total, got.q20, got.q30) == (s.sum, s.q20, s.q30)
        assert np.array_equal(got.qual_hist, want.quality)
        if base is not None:
            assert np.array_equal(got.nuc_hist, _nuc_ref(base, off))
    return want


def _nuc_ref(base, off):
    """Nucleotide[5][512] per Rgzfastq_uniq.c:50-57,97-108 (numpy restatement for the test)."""
    lut = np.zeros(256, np.int64)
    for ch, v in ((b"tTuU", 0), (b"cC", 1), (b"aA", 2), (b"gG", 3), (b".N", 4)):
        for c in ch:
            lut[c] = v
    out = np.zeros((5, 512), np.uint64)
    lens = np.diff(off.astype(np.int64))
    pos = np.arange(int(off[-1] - off[0])) - np.repeat(off[:-1].astype(np.int64) - int(off[0]), lens)
    np.add.at(out, (lut[base[int(off[0]):int(off[-1])]], pos), 1)
    return out


def test_appendix_a1_batch(ctx):
    # the five reads of SURVEY Appendix A.1 as a split batch
    quals = [b'@"9<G!=2/F', b"B/=@D/7//>", b"43F@A:F#?0:;", b"#4HFF:++A/!-CD/", b"BD.<$?8ED-A;"]
    seqs = [b"NAGATTTTCA", b"GAAANATCTA", b"ATNACGAGNTNC", b"CGNGATNACNTGTAT", b"NGNGTGNNATNC"]
    qual = np.frombuffer(b"".join(quals), np.uint8)
    base = np.frombuffer(b"".join(seqs), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(q) for q in quals])]).astype(np.uint64)
    want = _check(ctx, qual, off, base)
    s = want.summary()
    assert (s.reads, s.sum, s.min_len, s.max_len) == (5, 59, 10, 15)
    assert "%.3f %.3f" % (100.0 * s.q20 / s.sum, 100.0 * s.q30 / s.sum) == "61.017 38.983"


@pytest.mark.parametrize("n,lo,hi", [(1, 1, 1), (1, 511, 511), (7, 0, 3), (64, 150, 150), (65, 150, 150),
                                     (1000, 100, 100), (1025, 30, 151), (5000, 1, 511), (20000, 150, 150),
                                     (70000, 36, 36), (3000, 0, 0), (3000, 300, 300), (1500, 511, 511),
                                     (4000, 16, 16), (4000, 15, 15), (2500, 255, 257), (66000, 250, 250)])
def test_synthetic_batches(ctx, n, lo, hi):
    seq, qual, off = orc.synth_soa(n * 31 + lo, 0, n, lo, hi)
    _check(ctx, qual, off, seq)


@pytest.mark.parametrize("big", ["2", "4", "7"])
def test_turns_of_several_chunks(request, big):
    """K1L takes `big` chunks of 4096 records per workgroup turn when the batch is large; a turn whose records all have one
    length is streamed as one chunk, any other turn chunk by chunk.  Forced here on small batches: one-length turns, a turn
    with a single odd record in its last chunk, a ragged turn between one-length turns, a batch that ends inside a turn with
    a partial last group (length 150 = 18 x 8 + 6: the batch's last item cannot be loaded whole), lengths off every alignment."""
    from conftest import in_hooks_build
    if in_hooks_build(request, {"HPN_K1L_BIG": big}):     # (the switch lives in the test-hooks library: host/knobs.hpp)
        return
    ctx = request.getfixturevalue("ctx")
    for L, n_tail in ((150, 4096 * 3 + 17), (151, 5000), (64, 4096), (255, 777)):
        parts = [orc.synth_soa(11 + L, 0, 4096 * 8, L, L), orc.synth_soa(12 + L, 0, 1, L + 1, L + 1), orc.synth_soa(13 + L, 0, 4096 * 5 - 1, L, L),
                 orc.synth_soa(14 + L, 0, 3000, 20, 300), orc.synth_soa(15 + L, 0, n_tail, L, L)]
        seq = np.concatenate([p[0] for p in parts])
        qual = np.concatenate([p[1] for p in parts])
        lens = np.concatenate([np.diff(p[2].astype(np.int64)) for p in parts])
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        _check(ctx, qual, off, seq)


def test_empty_batch(ctx):
    got = ctx.fastq_tally(np.zeros(0, np.uint8), np.zeros(1, np.uint64), qual_hist=True)
    assert got.total == 0 and got.seqlen.sum() == 0 and got.qual_hist.sum() == 0
    got = ctx.fastq_tally(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert got.total == 0 and got.seqlen.sum() == 0


@pytest.mark.parametrize("skip", [1, 2, 3, 5, 17])
def test_sub_batches_at_odd_offsets(ctx, skip):
    # off[0] != 0 and not 16-byte aligned: the batch is a window into a larger array
    seq, qual, off = orc.synth_soa(77, 0, 400, 20, 130)
    sub = off[skip:-skip]
    rc, want = orc.count_soa(qual, sub)
    s = want.summary()
    for full in (False, True):
        got = ctx.fastq_tally(qual, sub, qual_hist=full)
        assert np.array_equal(got.seqlen, want.seqlen)
        assert (got.total, got.q20, got.q30) == (s.sum, s.q20, s.q30)
        if full:
            assert np.array_equal(got.qual_hist, want.quality)


def test_accumulates_like_count_read(ctx):
    # callee only adds into caller-owned accumulators (fastq_count_kthread.c:116)
    import highperformancengs_amd as hp
    seq, qual, off = orc.synth_soa(5, 0, 3000, 50, 150)
    acc = hp.TallyResult(qual_hist=True)
    cut = 1234
    ctx.fastq_tally(qual, off[:cut + 1], acc=acc)
    ctx.fastq_tally(qual, off[cut:], acc=acc)
    rc, want = orc.count_soa(qual, off)
    assert np.array_equal(acc.seqlen, want.seqlen) and np.array_equal(acc.qual_hist, want.quality)


def test_extreme_quality_bytes(ctx):
    # bytes 0, 52/53, 62/63 and 127 sit on the thresholds of statQ(…,53,…,63,…)
    vals = np.array([0, 1, 52, 53, 54, 62, 63, 64, 126, 127], np.uint8)
    qual = np.tile(vals, 300)
    off = np.arange(0, len(qual) + 1, 30, dtype=np.uint64)
    _check(ctx, qual, off)


@pytest.mark.parametrize("full", [False, True])
def test_domain_errors_are_reported(ctx, full):
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    qual = np.full(2000, 40, np.uint8)
    # read of length 512: SeqLen[512] overrun in the reference
    with pytest.raises(hp.HpnError) as e:
        ctx.fastq_tally(qual, np.array([0, 100, 612, 700], np.uint64), qual_hist=full)
    assert e.value.status == _lib.E_DOMAIN
    # quality byte >= 128: Quality[] row overrun in the reference
    q2 = qual.copy()
    q2[777] = 200
    with pytest.raises(hp.HpnError) as e:
        ctx.fastq_tally(q2, np.arange(0, 2001, 100, dtype=np.uint64), qual_hist=full)
    assert e.value.status == _lib.E_DOMAIN
    # the context is clean afterwards
    got = ctx.fastq_tally(qual, np.arange(0, 2001, 100, dtype=np.uint64), qual_hist=full)
    assert got.total == 2000 and got.seqlen[100] == 20


def test_device_generator_matches_cpu_generator(ctx):
    import torch
    n, length = 5000, 150
    dq = torch.empty(n * length, dtype=torch.uint8, device="cuda")
    db = torch.empty(n * length, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(4242, 1000, n, length, dq, db, do)
    ctx.sync()
    seq, qual, off = orc.synth_soa(4242, 1000, n, length, length)
    assert np.array_equal(dq.cpu().numpy(), qual)
    assert np.array_equal(db.cpu().numpy(), seq)
    assert np.array_equal(do.cpu().numpy().astype(np.uint64), off)


def test_device_resident_million_reads(ctx):
    """BASELINE configs[0]-sized batch (1e6 x 100 bp) resident in HBM vs the oracle."""
    import torch
    n, length = 1_000_000, 100
    dq = torch.empty(n * length, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(12345, 0, n, length, dq, None, do)
    ctx.fastq_tally_dev(dq, do, n)
    fast = ctx.fastq_tally_fetch()
    ctx.fastq_tally_dev(dq, do, n, flags=1)
    full = ctx.fastq_tally_fetch(qual_hist=True)
    seq, qual, off = orc.synth_soa(12345, 0, n, length, length)
    rc, want = orc.count_soa(qual, off)
    s = want.summary()
    for got in (fast, full):
        assert np.array_equal(got.seqlen, want.seqlen)
        assert (got.total, got.q20, got.q30) == (s.sum, s.q20, s.q30)
    assert np.array_equal(full.qual_hist, want.quality)
    # closed form of the generator: Phred uniform 2..41
    assert abs(100.0 * s.q20 / s.sum - 55.0) < 0.05 and abs(100.0 * s.q30 / s.sum - 30.0) < 0.05


def test_large_resident_batch_properties(ctx):
    """2e8 reads x 150 bp (30 GB) in HBM: size-independent checks.
    total = off[n]-off[0]; every read has length 150; tally(A)+tally(B) = tally(A u B);
    a window of it equals the oracle on the regenerated window."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    n, length = 200_000_000, 150
    if free < n * length * 1.2 + (n + 1) * 8:
        n = int(free * 0.5 / (length + 8))
    dq = torch.empty(n * length, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(99, 0, n, length, dq, None, do)
    ctx.fastq_tally_dev(dq, do, n)
    whole = ctx.fastq_tally_fetch()
    assert whole.total == n * length and whole.seqlen[length] == n and whole.seqlen.sum() == n
    cut = n // 3 + 7
    ctx.fastq_tally_dev(dq, do, cut)
    a = ctx.fastq_tally_fetch()
    ctx.fastq_tally_dev(dq, do[cut:], n - cut)
    b = ctx.fastq_tally_fetch()
    assert a.total + b.total == whole.total and a.q20 + b.q20 == whole.q20 and a.q30 + b.q30 == whole.q30
    # window [w0, w0+m) vs oracle
    w0, m = n // 2 + 3, 200_000
    ctx.fastq_tally_dev(dq, do[w0:], m, flags=1)
    win = ctx.fastq_tally_fetch(qual_hist=True)
    seq, qual, off = orc.synth_soa(99, w0, m, length, length)
    rc, want = orc.count_soa(qual, off)
    s = want.summary()
    assert (win.total, win.q20, win.q30) == (s.sum, s.q20, s.q30)
    assert np.array_equal(win.qual_hist, want.quality)
    assert abs(whole.q20 / whole.total - 0.55) < 1e-4 and abs(whole.q30 / whole.total - 0.30) < 1e-4
