"""GPU: randomized shapes against the oracle -- ragged lengths, odd batch windows, mixed CIGARs.
Deterministic seeds; every comparison is bit-exact.  HPN_FUZZ_X=k runs k times as many seeds (a one-off campaign)."""
import os

import numpy as np
import pytest

import orc
from bam_synth import make_soa

pytestmark = pytest.mark.gpu
_X = max(1, int(os.environ.get("HPN_FUZZ_X", "1")))


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _ragged(rng, n, mode):
    if mode == 0:      # uniform random lengths
        lens = rng.integers(0, 512, n)
    elif mode == 1:    # mostly one length with a few outliers (uniform chunks broken by single reads)
        lens = np.full(n, int(rng.integers(16, 300)))
        lens[rng.integers(0, n, max(1, n // 997))] = rng.integers(0, 512, max(1, n // 997))
    elif mode == 2:    # runs of equal lengths (several uniform chunks of different lengths)
        lens = np.repeat(rng.integers(1, 512, n // 4096 + 2), 4096)[:n]
    else:              # tiny reads
        lens = rng.integers(0, 20, n)
    off = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    tot = int(off[-1])
    qual = rng.integers(0, 128, max(tot, 1)).astype(np.uint8)[:tot]
    base = np.frombuffer(b"ACGTNacgtn.XU", np.uint8)[rng.integers(0, 13, max(tot, 1))][:tot]
    return base, qual, off


@pytest.mark.parametrize("seed", range(12 * _X))
def test_tally_fuzz(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 30000))
    base, qual, off = _ragged(rng, n, seed % 4)
    lo = int(rng.integers(0, min(n, 50)))
    hi = n - int(rng.integers(0, min(n - lo, 50)))
    sub = off[lo:hi + 1]
    rc, want = orc.count_soa(qual, sub)
    assert rc == 0
    s = want.summary()
    fast = ctx.fastq_tally(qual, sub)
    full = ctx.fastq_tally(qual, sub, base=base, qual_hist=True, nuc_hist=True)
    for got in (fast, full):
        assert np.array_equal(got.seqlen, want.seqlen) and (got.total, got.q20, got.q30) == (s.sum, s.q20, s.q30)
    assert np.array_equal(full.qual_hist, want.quality)
    assert int(full.nuc_hist.sum()) == s.sum
    # N/'.' column and G+C columns from the nucleotide histogram agree with a direct count
    seg = base[int(sub[0]):int(sub[-1])]
    assert int(full.nuc_hist[4].sum()) == int(np.isin(seg, np.frombuffer(b"N.", np.uint8)).sum())
    assert int(full.nuc_hist[1].sum() + full.nuc_hist[3].sum()) == int(np.isin(seg, np.frombuffer(b"CcGg", np.uint8)).sum())


@pytest.mark.parametrize("seed", range(8 * _X))
def test_trim_fuzz(ctx, seed):
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(1, 20000))
    base, qual, off = _ragged(rng, n, seed % 4)
    S = int(rng.integers(0, 200))
    E = S + int(rng.integers(0, 300))
    rc, wseq, wqual, woff = orc.trim_soa(base, qual, off, S, E)
    gseq, gqual, goff = ctx.fastq_trim(base, qual, off, S, E)
    assert np.array_equal(goff, woff) and np.array_equal(gseq, wseq) and np.array_equal(gqual, wqual)


@pytest.mark.parametrize("seed", range(6 * _X))
def test_depth_window_fuzz(ctx, seed):
    rng = np.random.default_rng(3000 + seed)
    refs = [("a", int(rng.integers(1000, 3_000_000))), ("b", int(rng.integers(300, 200_000))), ("c", 17)]
    n = int(rng.integers(0, 60000))
    soa = make_soa(n, refs, 4000 + seed, sort=bool(seed % 2 == 0), max_start_frac=0.999)
    W = int(rng.choice([1, 13, 100, 1000, 4096, 20000, 5_000_000]))
    for tid, (name, tlen) in enumerate(refs):
        if tlen // W + 1 > 1_000_000:
            continue
        runs, win = ctx.depth_target(soa, tid, tlen, W, 0x704 if seed % 3 else 0x4)
        rc, wruns, wbins = orc.depth_target(soa, tid, W, 0x704 if seed % 3 else 0x4)
        assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins)
    Ww = int(rng.choice([50, 1000, 20000]))
    if all(t // Ww + 1 <= 65536 for _, t in refs):
        rc, off, wb, wg, wl, wt, wn = orc.window_counts(soa, Ww)
        if rc == 0:
            bins, gc, ln, touched, nc = ctx.window_counts(soa, off, Ww)
            assert np.array_equal(bins, wb) and np.array_equal(gc, wg) and np.array_equal(ln, wl) and nc == wn
