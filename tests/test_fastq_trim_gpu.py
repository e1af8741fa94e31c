"""GPU parity: hpn_fastq_trim (HIP, through the C ABI) vs the oracle's cut and the
reference fastq_trim's golden output.  Bit-exact (bytes and offsets)."""
import numpy as np
import pytest

import orc
from conftest import expected

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _check(ctx, seq, qual, off, S, E):
    rc, wseq, wqual, woff = orc.trim_soa(seq, qual, off, S, E)
    assert rc == 0
    gseq, gqual, goff = ctx.fastq_trim(seq, qual, off, S, E)
    assert np.array_equal(goff, woff)
    assert np.array_equal(gseq, wseq) and np.array_equal(gqual, wqual)
    return gseq, gqual, goff


def test_appendix_a1_against_reference_text(ctx):
    # reference: fastq_trim -i t.fq -s 2 -e 8 (tests/golden/expected/trim_a1)
    quals = [b'@"9<G!=2/F', b"B/=@D/7//>", b"43F@A:F#?0:;", b"#4HFF:++A/!-CD/", b"BD.<$?8ED-A;"]
    seqs = [b"NAGATTTTCA", b"GAAANATCTA", b"ATNACGAGNTNC", b"CGNGATNACNTGTAT", b"NGNGTGNNATNC"]
    seq = np.frombuffer(b"".join(seqs), np.uint8)
    qual = np.frombuffer(b"".join(quals), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(q) for q in quals])]).astype(np.uint64)
    gseq, gqual, goff = _check(ctx, seq, qual, off, 2, 8)
    text = b""
    for i in range(5):
        a, b = int(goff[i]), int(goff[i + 1])
        text += b"@r%d desc\n%s\n+\n%s\n" % (i, gseq[a:b].tobytes(), gqual[a:b].tobytes())
    assert text == expected("trim_a1")


def test_golden_synthetic_against_reference_text(ctx):
    # reference: fastq_trim -i syn_var_b.fq.gz -s 5 -e 80
    seq, qual, off = orc.synth_soa(12345, 1500, 1500, 30, 151)
    gseq, gqual, goff = _check(ctx, seq, qual, off, 5, 80)
    want = expected("trim_syn_var").split(b"\n")
    for i in range(1500):
        a, b = int(goff[i]), int(goff[i + 1])
        assert want[4 * i + 1] == gseq[a:b].tobytes() and want[4 * i + 3] == gqual[a:b].tobytes()


@pytest.mark.parametrize("n,lo,hi,S,E", [(1, 10, 10, 0, 400), (3, 0, 0, 0, 5), (64, 150, 150, 5, 140),
                                         (65, 150, 150, 0, 150), (1000, 1, 300, 7, 7), (1024, 30, 151, 40, 100),
                                         (1025, 30, 151, 0, 1), (5000, 0, 511, 200, 400), (4097, 100, 100, 99, 100),
                                         (20000, 36, 36, 36, 40), (3000, 8, 200, 150, 1000)])
def test_synthetic(ctx, n, lo, hi, S, E):
    seq, qual, off = orc.synth_soa(n * 7 + S, 0, n, lo, hi)
    _check(ctx, seq, qual, off, S, E)


def test_empty_and_odd_windows(ctx):
    gseq, gqual, goff = ctx.fastq_trim(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64), 0, 10)
    assert len(gseq) == 0 and goff.tolist() == [0]
    seq, qual, off = orc.synth_soa(3, 0, 300, 20, 90)
    for skip in (1, 3, 10):
        _check(ctx, seq, qual, off[skip:-skip], 4, 33)


def test_domain_error(ctx):
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    seq, qual, off = orc.synth_soa(3, 0, 10, 20, 90)
    for S, E in ((5, 3), (-1, 4)):
        with pytest.raises(hp.HpnError) as e:
            ctx.fastq_trim(seq, qual, off, S, E)
        assert e.value.status == _lib.E_DOMAIN


def test_device_resident_closed_form(ctx):
    """BASELINE configs[2] shape (150 bp, -s 5 -e 140) on 2e7 resident reads: every output
    read has 135 bytes, out_off is 135*i, and a window equals the oracle's cut."""
    import torch
    n, L, S, E = 20_000_000, 150, 5, 140
    dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(31337, 0, n, L, dq, db, do)
    oq = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
    ob = torch.empty(n * (E - S), dtype=torch.uint8, device="cuda")
    oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.fastq_trim_dev(db, dq, do, n, S, E, ob, oq, oo)
    ctx.sync()
    assert torch.equal(oo, torch.arange(n + 1, device="cuda", dtype=torch.int64) * (E - S))
    # the cut is a strided view of the input: compare everything on the device
    assert torch.equal(oq.view(n, E - S), dq.view(n, L)[:, S:E])
    assert torch.equal(ob.view(n, E - S), db.view(n, L)[:, S:E])
    w0, m = 12_345_678, 5000
    seq, qual, off = orc.synth_soa(31337, w0, m, L, L)
    rc, wseq, wqual, woff = orc.trim_soa(seq, qual, off, S, E)
    assert np.array_equal(oq[w0 * (E - S):(w0 + m) * (E - S)].cpu().numpy(), wqual)
    assert np.array_equal(ob[w0 * (E - S):(w0 + m) * (E - S)].cpu().numpy(), wseq)


# ---- extension: quality-threshold trim points (no reference counterpart) ------------

@pytest.mark.parametrize("n,lo,hi,T", [(1, 10, 10, 53), (500, 0, 5, 40), (3000, 30, 151, 53), (3000, 30, 151, 63),
                                       (2000, 150, 150, 75), (2000, 150, 150, 0), (4000, 1, 511, 60), (70, 64, 64, 50),
                                       (70, 65, 129, 70)])
def test_qtrim_points_and_cut(ctx, n, lo, hi, T):
    seq, qual, off = orc.synth_soa(n + T, 0, n, lo, hi)
    wb, we = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    orc.lib().orc_qtrim_points(qual if len(qual) else np.zeros(1, np.uint8), off, n, T, wb, we)
    gb, ge = ctx.fastq_qtrim_points(qual, off, T)
    assert np.array_equal(gb, wb) and np.array_equal(ge, we)
    cap = max(len(qual), 1)
    wseq, wqual, woff = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(n + 1, np.uint64)
    orc.lib().orc_trim_points_soa(seq if len(seq) else np.zeros(1, np.uint8), qual if len(qual) else np.zeros(1, np.uint8),
                                  off, n, wb, we, wseq, wqual, woff)
    gseq, gqual, goff = ctx.fastq_trim_points(seq, qual, off, gb, ge)
    tot = int(woff[-1])
    assert np.array_equal(goff, woff) and np.array_equal(gseq, wseq[:tot]) and np.array_equal(gqual, wqual[:tot])
    # every kept base run starts and ends with a base at or above the threshold
    for i in range(min(n, 200)):
        a, b = int(goff[i]), int(goff[i + 1])
        if b > a:
            assert gqual[a] >= T and gqual[b - 1] >= T
