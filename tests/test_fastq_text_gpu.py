"""GPU parity: the raw-text front end (hpn_fastq_text_count / hpn_fastq_text_trim: newline
index, record validation and scan, gather / formatter on the device) against the oracle's
4 x gzgets stream loops, which test_oracle_golden.py pins to the reference binaries.

Contract under test: on REGULAR text the device framing gives exactly what the gzgets loop
gives, for any chunking of the stream; anything else is reported as irregular (and the
tools then take the exact host framer) -- never a silently different tally."""
import gzip
import os

import numpy as np
import pytest

import orc
from conftest import expected, golden_path

pytestmark = pytest.mark.gpu

FASTQS = ["t.fq", "t.fq.gz", "empty.fq", "nonl.fq", "crlf.fq", "multi.fq.gz", "short.fq", "len0.fq", "allzero.fq",
          "trunc.fq", "longname.fq", "syn_var_a.fq", "syn_var_b.fq.gz", "syn_100.fq.gz", "stale.fq"]
# files the fast path must accept (so that it cannot pass by always bailing out)
REGULAR = {"t.fq", "t.fq.gz", "empty.fq", "crlf.fq", "multi.fq.gz", "syn_var_a.fq", "syn_var_b.fq.gz", "syn_100.fq.gz"}


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _text(path):
    raw = open(path, "rb").read()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


def _chunks(text, size):
    if size is None or size >= len(text):
        return [text]
    return [text[i:i + size] for i in range(0, len(text), size)]


def _count(ctx, text, size, tail_call=False):
    """-> (TallyResult or None if irregular, flags, n_records)"""
    from highperformancengs_amd import _lib
    ctx.text_begin()
    parts = _chunks(text, size)
    if tail_call:
        parts = parts + [b""]  # the tools deliver EOF as an empty last chunk
    n = 0
    for i, p in enumerate(parts):
        info = ctx.text_count(p, last=(i == len(parts) - 1), flags=_lib.TALLY_QUAL_HIST)
        if info.irregular:
            try:
                ctx.fastq_tally_fetch()  # drop what earlier chunks added
            except Exception:
                pass
            return None, info.irregular, n
        n += info.n_records
    return ctx.fastq_tally_fetch(qual_hist=True), 0, n


def _assert_counts(res, want):
    assert np.array_equal(res.seqlen, want.seqlen)
    assert np.array_equal(res.qual_hist, want.quality)
    assert res.total == int(want.quality.sum())
    assert res.q20 == int(want.quality[53:].sum()) and res.q30 == int(want.quality[63:].sum())


@pytest.mark.parametrize("name", FASTQS)
def test_count_golden_files_any_chunking(ctx, name):
    path = golden_path("fastq", name)
    text = _text(path)
    rc, want = orc.count_stream(path)
    res, flags, n = _count(ctx, text, None)
    if name in REGULAR:
        assert flags == 0, f"{name}: fast path refused (flags {flags})"
    if res is None:
        return
    assert rc == 0
    _assert_counts(res, want)
    assert n == int(want.seqlen.sum())
    for size, tail in ((1000, False), (97, True), (16384, True), (31, False)):
        if len(text) / size > 3000:
            continue
        res2, flags2, n2 = _count(ctx, text, size, tail)
        assert flags2 == 0 and n2 == n, (name, size, flags2)
        _assert_counts(res2, want)


def test_count_last_newline_missing(ctx):
    """A final quality line without '\\n': the reference's strlen-1 only bites line 2 (fastq_count.c:114)."""
    text = _text(golden_path("fastq", "syn_var_a.fq"))
    assert text.endswith(b"\n")
    rc, want = orc.count_stream(golden_path("fastq", "syn_var_a.fq"))
    for size in (None, 4096, 333):
        res, flags, _ = _count(ctx, text[:-1], size)
        assert flags == 0
        _assert_counts(res, want)


TRIMS = [("t.fq", 2, 8), ("t.fq.gz", 0, 400), ("crlf.fq", 1, 3), ("syn_var_b.fq.gz", 5, 80), ("multi.fq.gz", 4, 9),
         ("empty.fq", 0, 9), ("syn_var_a.fq", 0, 0), ("syn_var_a.fq", 149, 150), ("syn_100.fq.gz", 10, 90),
         ("nonl.fq", 0, 10), ("short.fq", 3, 6), ("trunc.fq", 0, 50), ("longname.fq", 0, 50), ("len0.fq", 0, 5),
         ("stale.fq", 0, 10), ("stale.fq", 8, 40), ("t.fq", 12, 30), ("t.fq", 10, 30)]


def _trim(ctx, text, S, E, size, tail_call=False):
    ctx.text_begin()
    parts = _chunks(text, size)
    if tail_call:
        parts = parts + [b""]
    out, n = [], 0
    for i, p in enumerate(parts):
        o, info = ctx.text_trim(p, S, E, last=(i == len(parts) - 1))
        if info.irregular:
            return None, info.irregular, n
        out.append(o)
        n += info.n_records
    return b"".join(out), 0, n


@pytest.mark.parametrize("name,S,E", TRIMS)
def test_trim_golden_files_any_chunking(ctx, name, S, E):
    path = golden_path("fastq", name)
    text = _text(path)
    rc, want, nwant = orc.trim_stream(path, S, E)
    got, flags, n = _trim(ctx, text, S, E, None)
    shortest = min((len(s.rstrip(b"\r")) for s in text.split(b"\n")[1::4]), default=0)
    if name in REGULAR and S <= shortest:
        assert flags == 0, f"{name}: fast path refused (flags {flags})"
    if S > shortest:  # the reference copies stale buffer bytes there: host framer only
        assert flags != 0, name
    if got is None:
        return
    assert rc == 0 and got == want and n == nwant
    for size, tail in ((1000, True), (97, False), (50000, True)):
        if len(text) / size > 3000:
            continue
        got2, flags2, n2 = _trim(ctx, text, S, E, size, tail)
        assert flags2 == 0 and got2 == want and n2 == nwant, (name, size)


def test_trim_reference_golden_text(ctx):
    # reference: fastq_trim -i t.fq -s 2 -e 8 and fastq_trim -i syn_var_b.fq.gz -s 5 -e 80
    got, flags, _ = _trim(ctx, _text(golden_path("fastq", "t.fq")), 2, 8, None)
    assert flags == 0 and got == expected("trim_a1")
    got, flags, _ = _trim(ctx, _text(golden_path("fastq", "syn_var_b.fq.gz")), 5, 80, 7777)
    assert flags == 0 and got == expected("trim_syn_var")


def _random_fastq(rng, n, lo, hi, crlf=False):
    recs = []
    for i in range(n):
        l = int(rng.integers(lo, hi + 1))
        name = b"@r%d " % i + bytes(rng.integers(48, 123, int(rng.integers(0, 40)), dtype=np.uint8))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), l))
        qual = bytes(rng.integers(33, 75, l, dtype=np.uint8))
        plus = b"+" + (name[1:] if rng.random() < 0.3 else b"")
        eol = b"\r\n" if crlf else b"\n"
        recs.append(name + eol + seq + eol + plus + eol + qual + eol)
    return recs


@pytest.mark.parametrize("seed,n,lo,hi,crlf", [(1, 2000, 0, 300, False), (2, 5000, 30, 151, False), (3, 300, 1, 511, False),
                                               (4, 1000, 20, 100, True), (5, 70000, 100, 100, False)])
def test_count_and_trim_random_regular_text(ctx, tmp_path, seed, n, lo, hi, crlf):
    rng = np.random.default_rng(seed)
    text = b"".join(_random_fastq(rng, n, lo, hi, crlf))
    p = tmp_path / "r.fq"
    p.write_bytes(text)
    rc, want = orc.count_stream(str(p))
    assert rc == 0
    for size in (None, int(rng.integers(5000, 200000)), 1 << 20):
        res, flags, nrec = _count(ctx, text, size, tail_call=bool(size))
        assert flags == 0 and nrec == n
        _assert_counts(res, want)
    # S beyond a read's end is outside the tools' domain (the reference copies stale buffer bytes)
    S, E = min(int(rng.integers(0, 40)), lo), int(rng.integers(40, 200))
    rc, wtext, nw = orc.trim_stream(str(p), S, E)
    for size in (None, int(rng.integers(5000, 200000))):
        got, flags, nrec = _trim(ctx, text, S, E, size)
        assert flags == 0 and nrec == n and got == wtext


def _mutate(rng, text):
    """One irregularity somewhere in the text."""
    kind = int(rng.integers(0, 7))
    b = bytearray(text)
    nls = np.flatnonzero(np.frombuffer(text, np.uint8) == 10)
    if kind == 0 and len(nls):      # drop a newline: two lines merge, framing shifts
        del b[int(rng.choice(nls))]
    elif kind == 1:                 # NUL byte
        b[int(rng.integers(0, len(b)))] = 0
    elif kind == 2:                 # a line longer than the gzgets buffer
        i = int(rng.integers(0, len(b)))
        b[i:i] = b"A" * int(rng.integers(1023, 3000))
    elif kind == 3:                 # truncate anywhere
        del b[int(rng.integers(1, len(b))):]
    elif kind == 4 and len(nls):    # extra newline: blank line, framing shifts
        b[int(rng.choice(nls)):int(rng.choice(nls))] = b""
        b.insert(int(rng.integers(0, len(b))), 10)
    elif kind == 5 and len(nls) > 8:  # shorten one quality line
        k = int(rng.integers(0, len(nls) // 4)) * 4 + 3
        if nls[k] - nls[k - 1] > 3:
            del b[int(nls[k]) - 2:int(nls[k])]
    else:                           # trailing garbage without newline
        b += b"@partial"
    return bytes(b)


@pytest.mark.parametrize("seed", range(24))
def test_irregular_text_is_detected_or_exact(ctx, tmp_path, seed):
    """Whatever the damage: either the fast path reports it, or its result IS the gzgets loop's."""
    rng = np.random.default_rng(1000 + seed)
    text = _mutate(rng, b"".join(_random_fastq(rng, int(rng.integers(50, 400)), 5, 200)))
    p = tmp_path / "m.fq"
    p.write_bytes(text)
    size = [None, 4096, 100][seed % 3]
    res, flags, _ = _count(ctx, text, size, tail_call=bool(seed & 1))
    if res is not None:
        rc, want = orc.count_stream(str(p))
        assert rc == 0
        _assert_counts(res, want)
    got, flags, n = _trim(ctx, text, 0, 60, size)  # S = 0: see the domain note above
    if got is not None:
        rc, wtext, nw = orc.trim_stream(str(p), 0, 60)
        assert got == wtext and n == nw


def test_state_errors(ctx):
    from highperformancengs_amd import HpnError
    ctx.text_begin()
    info = ctx.text_count(b"@a\nAC\n+\nII\n", last=True)
    assert info.irregular == 0 and info.n_records == 1 and info.n_bytes == 2
    with pytest.raises(HpnError):  # stream closed by last=True
        ctx.text_count(b"@a\nAC\n+\nII\n", last=True)
    ctx.fastq_tally_fetch()
    with pytest.raises(HpnError):
        ctx.text_begin()
        ctx.text_trim(b"@a\nAC\n+\nII\n", 5, 2, last=True)  # E < S


# ---- gzfastq_sample.c:214-225 count_read: four gzgets per record and i++ -------------------------
# The tool itself is unbuildable here (fastq-tools' common.c includes an autoconf-generated version.h), but its
# count_read is fastq_count's loop without the tally, so its i is the ReadCount column the reference's fastq_count
# prints for the same file: that column is golden.

@pytest.mark.parametrize("name", FASTQS)
def test_record_count_equals_the_reference_read_count(ctx, name, manifest):
    from conftest import expected
    case = next((k for k, c in manifest.items() if k.startswith("count_") and c["tool"] == "fastq_count" and len(c["inputs"]) == 1
                 and c["inputs"][0].split("/")[-1] == name and c["returncode"] == 0 and not c.get("stdin") and not c.get("stderr_usage")), None)
    if case is None:
        pytest.skip("no single-file fastq_count golden for " + name)
    row = [l for l in expected(case).decode().split("\n") if l and not l.startswith("#")][0].split("\t")
    want = int(row[1])
    text = _text(golden_path("fastq", name))
    for size in (None, 1000, 97):
        ctx.text_begin()
        parts = _chunks(text, size)
        n, irregular = 0, 0
        for i, p in enumerate(parts):
            info = ctx.text_records(p, last=(i == len(parts) - 1))
            if info.irregular:
                irregular = info.irregular
                break
            n += info.n_records
        if name in REGULAR:
            assert irregular == 0
        if not irregular:
            assert n == want, (name, size)
    # nothing was tallied by the framing-only call
    ctx.text_begin()
    ctx.text_records(b"@a\nAC\n+\nII\n", last=True)
    assert ctx.fastq_tally_fetch().total == 0


# ---- text that lies on the device already, framed where it lies (hpn_fastq_text_count_inplace, round 6) -----------------------
# The gzip route's batches are device text: the framer takes them in place (8192 writable bytes in front of the chunk for the
# carried bytes) instead of copying every chunk into a slot.  Same counts as the oracle's gzgets loop for any chunking, the same
# irregular reports, the text may be overwritten behind each call (the next batch's inflate does), and chunks framed in place
# and chunks handed over the ordinary way may follow each other.  (Reference loop: fastq_count.c:112-118.)
def _count_inplace(ctx, text, size, mix=False, clobber=False):
    import torch
    from highperformancengs_amd import _lib
    parts = _chunks(text, size) + [b""]
    ctx.text_begin()
    n = 0
    keep = []
    for i, p in enumerate(parts):
        last = i == len(parts) - 1
        if mix and i % 3 == 1:
            info = ctx.text_count(p, last=last, flags=_lib.TALLY_QUAL_HIST)
        else:
            buf = torch.full((8192 + len(p) + 64,), 0x41, dtype=torch.uint8, device="cuda")      # ('A's around it: never a newline)
            if len(p):
                buf[8192:8192 + len(p)] = torch.from_numpy(np.frombuffer(p, np.uint8).copy()).cuda()
            info = ctx.text_count_inplace(buf[8192:], len(p), last=last, flags=_lib.TALLY_QUAL_HIST)
            if clobber:
                buf.fill_(10)           # the caller writes over the text at once (here: newlines everywhere)
            keep.append(buf)
        if info.irregular:
            try:
                ctx.fastq_tally_fetch()
            except Exception:
                pass
            return None, info.irregular, n
        n += info.n_records
    return ctx.fastq_tally_fetch(qual_hist=True), 0, n


@pytest.mark.parametrize("size", [None, 31, 100, 4099, 70001])
@pytest.mark.parametrize("name", FASTQS)
def test_count_in_place_any_chunking(ctx, name, size):
    path = golden_path("fastq", name)
    text = _text(path)
    rc, want = orc.count_stream(path)
    want_res, want_flags, want_n = _count(ctx, text, size, tail_call=True)
    res, flags, n = _count_inplace(ctx, text, size)
    assert (flags != 0) == (want_flags != 0)
    if name in REGULAR:
        assert flags == 0, f"{name}: fast path refused (flags {flags})"
    if flags == 0:
        assert rc == 0 and n == want_n
        _assert_counts(res, want)


@pytest.mark.parametrize("mix,clobber", [(True, False), (False, True), (True, True)])
def test_count_in_place_mixed_with_copied_chunks_and_overwritten_text(ctx, mix, clobber):
    path = golden_path("fastq", "syn_var_a.fq")
    text = _text(path)
    rc, want = orc.count_stream(path)
    for size in (997, 65536):
        res, flags, n = _count_inplace(ctx, text, size, mix=mix, clobber=clobber)
        assert flags == 0 and rc == 0
        _assert_counts(res, want)


@pytest.mark.parametrize("seed", range(12))
def test_damage_deep_inside_a_large_text_is_detected_or_exact(ctx, tmp_path, seed):
    """The same contract on text of several tiles (k_text_lines takes whole words on every tile but a call's first and last:
    another code path than the small texts above reach) -- a NUL, a lost or an extra newline far from both ends."""
    rng = np.random.default_rng(7000 + seed)
    text = bytearray(b"".join(_random_fastq(rng, 6000, 40, 151)))      # ~1.2 MB: nine tiles of 128 KiB
    nls = np.flatnonzero(np.frombuffer(bytes(text), np.uint8) == 10)
    at = int(rng.integers(len(text) // 4, 3 * len(text) // 4))
    kind = seed % 4
    if kind == 0:
        text[at] = 0
    elif kind == 1:
        del text[int(nls[np.searchsorted(nls, at)])]
    elif kind == 2:
        text.insert(at, 10)
    else:   # nothing: the undamaged text must be taken by the fast path
        pass
    text = bytes(text)
    p = tmp_path / "d.fq"
    p.write_bytes(text)
    rc, want = orc.count_stream(str(p))
    for size in (None, 300_000):
        res, flags, _ = _count(ctx, text, size, tail_call=bool(size))
        if kind == 0:
            assert flags != 0, "a NUL byte inside a whole-word tile went unnoticed"
        if kind == 3:
            assert flags == 0
        if res is not None:
            assert rc == 0
            _assert_counts(res, want)
        res, flags, _ = _count_inplace(ctx, text, size)
        if kind == 0:
            assert flags != 0
        if res is not None:
            assert rc == 0
            _assert_counts(res, want)


@pytest.mark.parametrize("tiles,newline_at_end", [(1, True), (1, False), (3, True), (3, False)])
def test_text_that_ends_exactly_on_a_tile_boundary(ctx, tmp_path, tiles, newline_at_end):
    """In place the text starts 16-byte aligned, so nbytes = k * 128 KiB puts the stream's last byte on a tile's last byte:
    with a final newline, and with the virtual one an unterminated last line gets (position `end`, one past the tile)."""
    rng = np.random.default_rng(tiles * 2 + newline_at_end)
    want_len = tiles * 131072
    recs, size = [], 0
    while True:
        r = _random_fastq(rng, 1, 60, 120)[0]
        if size + len(r) > want_len - 400:
            break
        recs.append(r), (size := size + len(r))
    rest = want_len - size + (0 if newline_at_end else 1)      # the last record fills the rest exactly
    l = (rest - len(b"@last\n\n+\n\n")) // 2
    assert 1 <= l < 400
    pad = rest - 10 - 2 * l                                      # 0 or 1: a longer name
    recs.append(b"@last" + b"x" * pad + b"\n" + b"A" * l + b"\n+\n" + b"I" * l + b"\n")
    text = b"".join(recs)
    if not newline_at_end:
        text = text[:-1]
    assert len(text) == want_len
    p = tmp_path / "e.fq"
    p.write_bytes(text)
    rc, want = orc.count_stream(str(p))
    assert rc == 0
    for size in (None, 131072):
        res, flags, n = _count_inplace(ctx, text, size)
        assert flags == 0 and n == len(recs)
        _assert_counts(res, want)
    # ... and as ONE call that is also the stream's last (the helper above ends every stream with an empty call)
    import torch
    from highperformancengs_amd import _lib
    buf = torch.full((8192 + len(text) + 64,), 0x41, dtype=torch.uint8, device="cuda")
    buf[8192:8192 + len(text)] = torch.from_numpy(np.frombuffer(text, np.uint8).copy()).cuda()
    ctx.text_begin()
    info = ctx.text_count_inplace(buf[8192:], len(text), last=True, flags=_lib.TALLY_QUAL_HIST)
    assert info.irregular == 0 and info.n_records == len(recs)
    _assert_counts(ctx.fastq_tally_fetch(qual_hist=True), want)
