"""GPU parity: ONE FASTQ stream framed in pieces by several contexts (hpn_fastq_text_piece_lines / _count /
_trim, include/hpngs.h) -- the record-block sharding of a single input over the node's GPUs --
against the oracle's 4 x gzgets stream loops (fastq_count.c:112-118, fastq_trim.c:67-89).

Contract: for ANY cut points, the pieces' results add up to exactly what the serial loop gives (counts: the
sum of the contexts' vectors; trim: the concatenation in piece order), the only thing pieces exchange being
the number of lines in front of each; anything irregular is reported by some piece, never mis-framed."""
import gzip

import numpy as np
import pytest

import orc
from conftest import golden_path
from test_fastq_text_gpu import FASTQS, REGULAR, TRIMS, _assert_counts, _mutate, _random_fastq, _text

pytestmark = pytest.mark.gpu
TAIL = 4096


@pytest.fixture(scope="module")
def ctxs():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    cs = [hp.Context(0), hp.Context(0), hp.Context(0)]   # three lanes on the one device (HPN_NGPU's arrangement)
    yield cs
    for c in cs:
        c.close()


def _cuts(n, sizes):
    """Piece boundaries 0 = b0 < b1 < ... < n from a cycle of sizes."""
    b, i = [0], 0
    while b[-1] < n:
        b.append(min(n, b[-1] + sizes[i % len(sizes)]))
        i += 1
    if len(b) == 1:
        b.append(0)   # the empty stream is one (last) piece
    return b


def _pieces(text, cuts):
    n = len(text)
    for j in range(len(cuts) - 1):
        a, b = cuts[j], cuts[j + 1]
        head = 1 if j else 0
        last = j == len(cuts) - 2
        hi = n if last else min(n, b + TAIL)
        yield j, text[a - head:hi], head, b - a, last


def _run(ctxs, text, cuts, second, order=None):
    """Two passes like the tools' lanes, but serial: every piece's lines first on its lane (round-robin), the board of
    line counts, then every piece's second half.  A lane holds one piece at a time, so the halves of a piece run back
    to back; `order` shuffles which piece goes first (lanes race in the tools) -- the board is filled in stream order
    from a pre-pass, which is what the chain of waits amounts to."""
    pcs = list(_pieces(text, cuts))
    counts = []
    for j, t, head, own, last in pcs:
        pl = ctxs[j % len(ctxs)].text_piece_lines(t, head, own, last)
        if pl.irregular:
            return None, pl.irregular
        counts.append(int(pl.n_lines))
    before = np.concatenate([[0], np.cumsum(counts)])
    # the published counts are the '\n' in T[b_j - head, b_j+1 - 1)
    arr = np.frombuffer(text, np.uint8)
    for j, (_, _, head, own, last) in enumerate(pcs):
        if not last:
            a = cuts[j] - head
            assert counts[j] == int((arr[a:cuts[j + 1] - 1] == 10).sum()), j
    out = []
    for j in (order if order is not None else range(len(pcs))):
        _, t, head, own, last = pcs[j]
        c = ctxs[j % len(ctxs)]
        pl = c.text_piece_lines(t, head, own, last)
        assert pl.irregular == 0 and int(pl.n_lines) == counts[j]
        r = second(c, j, int(before[j]), len(t))
        if r is None:
            return None, -1
        out.append((j, r))
    return [r for _, r in sorted(out)], 0


def _count(ctxs, text, cuts, order=None):
    from highperformancengs_amd import _lib
    flags = []

    def second(c, j, before, cap):
        info = c.text_piece_count(before, _lib.TALLY_QUAL_HIST)
        if info.irregular:
            flags.append(info.irregular)
            return None
        return int(info.n_records)
    res, f = _run(ctxs, text, cuts, second, order)
    parts = []
    for c in ctxs:   # the host sum of the lanes' vectors (what LaneGroup::sum_into does without RCCL)
        try:
            parts.append(c.fastq_tally_fetch(qual_hist=True))
        except Exception:
            parts.append(None)
    if res is None:
        return None, (flags[0] if flags else f), 0
    from types import SimpleNamespace
    tot = SimpleNamespace(seqlen=sum(p.seqlen for p in parts), qual_hist=sum(p.qual_hist for p in parts),
                          total=sum(p.total for p in parts), q20=sum(p.q20 for p in parts), q30=sum(p.q30 for p in parts))
    return tot, 0, sum(res)


def _trim(ctxs, text, cuts, S, E, order=None):
    flags = []

    def second(c, j, before, cap):
        o, info = c.text_piece_trim(before, S, E, cap + 8192)
        if info.irregular:
            flags.append(info.irregular)
            return None
        return o, int(info.n_records)
    res, f = _run(ctxs, text, cuts, second, order)
    if res is None:
        return None, (flags[0] if flags else f), 0
    return b"".join(o for o, _ in res), 0, sum(n for _, n in res)


CUTS = [[1 << 30], [1000], [97, 333, 8192], [4096], [50000, 1, 2, 3], [31]]


@pytest.mark.parametrize("name", FASTQS)
def test_count_golden_files_any_cuts(ctxs, name):
    path = golden_path("fastq", name)
    text = _text(path)
    rc, want = orc.count_stream(path)
    whole, flags, n = _count(ctxs, text, _cuts(len(text), [1 << 30]))
    if name in REGULAR:
        assert flags == 0, f"{name}: the piece route refused regular text (flags {flags})"
    if whole is None:
        return
    assert rc == 0
    _assert_counts(whole, want)
    assert n == int(want.seqlen.sum())
    for sizes in CUTS[1:]:
        if len(text) / min(sizes) > 3000:
            continue
        res, flags2, n2 = _count(ctxs, text, _cuts(len(text), sizes))
        # a tiny LAST piece may leave the stream's last record to a piece that is not `last` (no virtual final newline
        # there): such cuts may be refused, never mis-framed
        if res is None:
            assert not text.endswith(b"\n") or name not in REGULAR, (name, sizes, flags2)
            continue
        assert n2 == n, (name, sizes)
        _assert_counts(res, want)


@pytest.mark.parametrize("name,S,E", TRIMS)
def test_trim_golden_files_any_cuts(ctxs, name, S, E):
    path = golden_path("fastq", name)
    text = _text(path)
    rc, want, nwant = orc.trim_stream(path, S, E)
    got, flags, n = _trim(ctxs, text, _cuts(len(text), [1 << 30]), S, E)
    shortest = min((len(s.rstrip(b"\r")) for s in text.split(b"\n")[1::4]), default=0)
    if name in REGULAR and S <= shortest:
        assert flags == 0, f"{name}: the piece route refused (flags {flags})"
    if S > shortest:
        assert flags != 0, name
    if got is None:
        return
    assert rc == 0 and got == want and n == nwant
    for sizes in ([1000], [97, 333, 8192], [50000, 7]):
        if len(text) / min(sizes) > 3000:
            continue
        got2, flags2, n2 = _trim(ctxs, text, _cuts(len(text), sizes), S, E)
        assert flags2 == 0 and got2 == want and n2 == nwant, (name, sizes)


@pytest.mark.parametrize("seed,n,lo,hi,crlf", [(11, 3000, 0, 300, False), (12, 5000, 30, 151, False), (13, 300, 1, 511, False),
                                               (14, 1000, 20, 100, True), (15, 40000, 100, 100, False)])
def test_random_regular_text_random_cuts_and_orders(ctxs, tmp_path, seed, n, lo, hi, crlf):
    rng = np.random.default_rng(seed)
    text = b"".join(_random_fastq(rng, n, lo, hi, crlf))
    p = tmp_path / "r.fq"
    p.write_bytes(text)
    rc, want = orc.count_stream(str(p))
    assert rc == 0
    S, E = min(int(rng.integers(0, 40)), lo), int(rng.integers(40, 200))
    rc, wtext, nw = orc.trim_stream(str(p), S, E)
    nls = np.flatnonzero(np.frombuffer(text, np.uint8) == 10)
    for trial in range(3):
        k = int(rng.integers(2, 40))
        # random cut points, some exactly behind a newline, some exactly on one, some at a record's first byte
        pts = set(int(x) for x in rng.integers(1, len(text), k))
        pts |= set(int(nls[int(i)]) + 1 for i in rng.integers(0, len(nls) - 1, 4))
        pts |= set(int(nls[int(i)]) for i in rng.integers(0, len(nls) - 1, 4))
        pts |= set(int(nls[4 * int(i) + 3]) + 1 for i in rng.integers(0, len(nls) // 4 - 1, 3))
        cuts = [0] + sorted(x for x in pts if 0 < x < len(text)) + [len(text)]
        order = list(rng.permutation(len(cuts) - 1))
        res, flags, nrec = _count(ctxs, text, cuts, order)
        assert flags == 0 and nrec == n, (trial, flags)
        _assert_counts(res, want)
        got, flags, nrec = _trim(ctxs, text, cuts, S, E, order)
        assert flags == 0 and nrec == n and got == wtext


def test_last_newline_missing(ctxs):
    """A final quality line without '\\n' (fastq_count.c:114's strlen - 1 only bites line 2): the last piece closes it."""
    text = _text(golden_path("fastq", "syn_var_a.fq"))
    rc, want = orc.count_stream(golden_path("fastq", "syn_var_a.fq"))
    for sizes in ([1 << 30], [60000], [8192, 5000]):
        res, flags, _ = _count(ctxs, text[:-1], _cuts(len(text) - 1, sizes))
        assert flags == 0
        _assert_counts(res, want)


@pytest.mark.parametrize("seed", range(24))
def test_irregular_text_is_detected_or_exact(ctxs, tmp_path, seed):
    rng = np.random.default_rng(3000 + seed)
    text = _mutate(rng, b"".join(_random_fastq(rng, int(rng.integers(50, 400)), 5, 200)))
    p = tmp_path / "m.fq"
    p.write_bytes(text)
    sizes = [[1 << 30], [4096], [700, 41]][seed % 3]
    res, flags, _ = _count(ctxs, text, _cuts(len(text), sizes))
    if res is not None:
        rc, want = orc.count_stream(str(p))
        assert rc == 0
        _assert_counts(res, want)
    got, flags, n = _trim(ctxs, text, _cuts(len(text), sizes), 0, 60)
    if got is not None:
        rc, wtext, nw = orc.trim_stream(str(p), 0, 60)
        assert got == wtext and n == nw


def test_records_of_maximal_lines_fit_the_tail(ctxs, tmp_path):
    """Lines of 1022 characters (the longest gzgets takes whole) make records of up to 4092 bytes: the 4 KiB tail holds
    the rest of any of them, wherever the cut falls."""
    rec = b"@" + b"n" * 1021 + b"\n" + b"A" * 511 + b"\n+" + b"p" * 1021 + b"\n" + b"I" * 1022 + b"\n"
    text = rec * 20
    p = tmp_path / "max.fq"
    p.write_bytes(text)
    rc, want = orc.count_stream(str(p))
    assert rc == 0
    for sizes in ([len(rec) * 3 + 1], [5000], [len(rec) * 5], [4097, 1]):
        res, flags, n = _count(ctxs, text, _cuts(len(text), sizes))
        assert flags == 0 and n == 20, sizes
        _assert_counts(res, want)


def test_state_errors(ctxs):
    from highperformancengs_amd import HpnError
    c = ctxs[0]
    with pytest.raises(HpnError):
        c.text_piece_count(0)                      # no piece pending
    with pytest.raises(HpnError):
        c.text_piece_lines(b"@a\nAC\n+\nII\n", 1, 20, True)   # head + own beyond the text
    pl = c.text_piece_lines(b"@a\nAC\n+\nII\n", 0, 11, True)
    assert pl.irregular == 0
    with pytest.raises(HpnError):
        c.text_piece_count(4)                      # a first piece has no lines in front of it
    pl = c.text_piece_lines(b"@a\nAC\n+\nII\n", 0, 11, True)
    info = c.text_piece_count(0)
    assert info.n_records == 1 and info.n_bytes == 2
    assert c.fastq_tally_fetch().total == 2
