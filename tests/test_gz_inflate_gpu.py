"""GPU: one DEFLATE stream inflated in stretches with the history unknown (hpn_gz_inflate_dev).

zlib is the oracle (it is what the reference's gzread calls).  The stretches are cut where the
bit position of a block boundary is known without a search: after a Z_FULL_FLUSH / Z_SYNC_FLUSH
marker the stream is byte-aligned at a block start.  (The tools find arbitrary block starts with
host/pgz_reader.hpp's trial decoder; tests/test_cli_gpu.py covers that route end to end.)"""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CHUNK = np.dtype([("in_off", "<u8"), ("end_bit", "<u8"), ("in_len", "<u4"), ("start_bit", "<u4")])
NONE = (1 << 64) - 1


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _stream(pieces, level=6, flush=zlib.Z_SYNC_FLUSH, strategy=zlib.Z_DEFAULT_STRATEGY):
    """raw deflate of the concatenated pieces; returns (bytes, [byte offset where piece i starts])."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    out, starts = b"", []
    for k, p in enumerate(pieces):
        starts.append(len(out))
        out += c.compress(p)
        out += c.flush(flush) if k + 1 < len(pieces) else c.flush()
    return out, starts


def _run(ctx, comp, starts, want, sym_cap=None, window=None, start_bits=None):
    n = len(starts)
    tab = np.zeros(n, CHUNK)
    for k in range(n):
        tab[k]["in_off"] = starts[k]
        tab[k]["start_bit"] = 0 if start_bits is None else start_bits[k]
        tab[k]["in_len"] = len(comp) - starts[k]
        tab[k]["end_bit"] = (starts[k + 1] - starts[k]) * 8 if k + 1 < n else NONE
    d_comp = torch.from_numpy(np.frombuffer(comp + bytes(128), np.uint8).copy()).cuda()
    d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
    cap = sym_cap or (max(len(w) for w in want) + 8 + 7) // 8 * 8
    total = sum(len(w) for w in want)
    d_text = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    d_wout = torch.zeros(32768, dtype=torch.uint8, device="cuda")
    d_win = torch.from_numpy(np.frombuffer(window, np.uint8).copy()).cuda() if window is not None else None
    info = ctx.gz_inflate_dev(d_comp, d_tab, n, cap, d_text, total + 64, d_win, d_wout)
    return info, d_text[:int(info.n_bytes)].cpu().numpy().tobytes() if not info.status else b"", d_wout.cpu().numpy().tobytes()


def _fastq(rng, n, L=100):
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, L))
    qual = rng.integers(35, 74, (n, L), dtype=np.uint8)
    return b"".join(b"@read%d/1\n%s\n+\n%s\n" % (i, seq[i].tobytes(), qual[i].tobytes()) for i in range(n))


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (5, zlib.Z_HUFFMAN_ONLY), (7, zlib.Z_RLE), (0, zlib.Z_DEFAULT_STRATEGY)])
def test_stretches_with_unknown_history_equal_zlib(ctx, level, strategy):
    rng = np.random.default_rng(level * 10 + strategy)
    text = _fastq(rng, 6000)
    cuts = sorted(int(x) for x in rng.integers(1, len(text), 11))
    pieces = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
    comp, starts = _stream(pieces, level, zlib.Z_SYNC_FLUSH, strategy)   # sync flush: later pieces refer back into earlier ones
    assert zlib.decompress(comp, -15) == text
    info, got, wout = _run(ctx, comp, starts, pieces)
    assert info.status == 0 and info.final_chunk == len(pieces)
    assert got == text
    assert wout == text[-32768:]
    assert info.end_bit == (len(comp) - starts[-1]) * 8   # the final block ends with the stream (byte-aligned)


def test_history_handed_in_and_out_across_calls(ctx):
    """A stream inflated in two calls: the first call's window_out is the second's window_in."""
    rng = np.random.default_rng(3)
    text = _fastq(rng, 4000)
    pieces = [text[i:i + 90000] for i in range(0, len(text), 90000)]
    comp, starts = _stream(pieces, 6, zlib.Z_SYNC_FLUSH)
    half = len(pieces) // 2
    # first call: stretches [0, half) -- the last one must stop at the start of stretch `half`
    n1 = starts[half]
    tab_want = pieces[:half]
    info1, got1, w1 = _run_partial(ctx, comp, starts[:half], n1, tab_want)
    assert info1.status == 0 and info1.final_chunk == 0 and got1 == b"".join(pieces[:half])
    info2, got2, w2 = _run(ctx, comp[n1:], [s - n1 for s in starts[half:]], pieces[half:], window=w1)
    assert info2.status == 0 and got2 == b"".join(pieces[half:])
    assert w2 == text[-32768:]


def _run_partial(ctx, comp, starts, stop, want):
    n = len(starts)
    tab = np.zeros(n, CHUNK)
    for k in range(n):
        tab[k]["in_off"] = starts[k]
        tab[k]["in_len"] = len(comp) - starts[k]
        tab[k]["end_bit"] = ((starts[k + 1] if k + 1 < n else stop) - starts[k]) * 8
    d_comp = torch.from_numpy(np.frombuffer(comp + bytes(128), np.uint8).copy()).cuda()
    d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
    cap = (max(len(w) for w in want) + 15) // 8 * 8
    total = sum(len(w) for w in want)
    d_text = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    d_wout = torch.zeros(32768, dtype=torch.uint8, device="cuda")
    info = ctx.gz_inflate_dev(d_comp, d_tab, n, cap, d_text, total + 64, None, d_wout)
    return info, d_text[:int(info.n_bytes)].cpu().numpy().tobytes(), d_wout.cpu().numpy().tobytes()


def test_payload_shapes(ctx):
    """Long matches, distance-1 runs, matches reaching exactly 32768 back, binary bytes, tiny and empty stretches."""
    rng = np.random.default_rng(9)
    block = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    pieces = [block, block, b"", b"x", bytes(100000), (b"ACGT" * 30000)[:77777], block[:100] * 50, rng.integers(0, 4, 50000, dtype=np.uint8).tobytes(),
              block]
    comp, starts = _stream(pieces, 9, zlib.Z_SYNC_FLUSH)
    info, got, wout = _run(ctx, comp, starts, pieces)
    text = b"".join(pieces)
    assert info.status == 0 and got == text and wout == text[-32768:]


def test_wrong_ends_and_damage_are_reported(ctx):
    rng = np.random.default_rng(4)
    text = _fastq(rng, 3000)
    pieces = [text[i:i + 100000] for i in range(0, len(text), 100000)]
    comp, starts = _stream(pieces, 6, zlib.Z_SYNC_FLUSH)
    # an end that is not a block boundary of the stream
    bad_starts = list(starts)
    bad_starts[2] += 3
    info, _, _ = _run(ctx, comp, bad_starts, pieces)
    assert info.status != 0 and info.bad_chunk in (1, 2)
    # too little room for the symbols
    info, _, _ = _run(ctx, comp, starts, pieces, sym_cap=4096)
    assert info.status != 0
    # flipped bits: either some stretch reports, or the text differs from the original exactly as zlib's does
    for trial in range(6):
        raw = bytearray(comp)
        pos = int(rng.integers(0, len(raw)))
        raw[pos] ^= 1 << int(rng.integers(0, 8))
        info, got, _ = _run(ctx, bytes(raw), starts, pieces, sym_cap=(len(text) + 15) // 8 * 8)
        try:
            want = zlib.decompress(bytes(raw), -15)
        except zlib.error:
            want = None
        if info.status == 0:
            assert want is not None and got == want, trial


def test_final_block_inside_a_bounded_stretch_is_reported(ctx):
    """Two deflate streams back to back (what `cat a.gz b.gz` holds between its members' headers): a stretch that was given
    an end but meets a final block stops before that end, so the next stretch's start is not proven -- status 22, whatever
    the bytes behind the first stream look like."""
    rng = np.random.default_rng(22)
    a, b = _fastq(rng, 1500), _fastq(rng, 1500)
    ca, sa = _stream([a[:60000], a[60000:]], 6, zlib.Z_SYNC_FLUSH)
    cb, sb = _stream([b], 6, zlib.Z_SYNC_FLUSH)
    comp = ca + cb
    # stretch 1 is told to end where the second stream starts; it ends earlier, at stream a's final block ... exactly there
    # in bytes, but through a final block, which a stretch with an end must not contain
    starts = [sa[0], sa[1], len(ca)]
    info, got, _ = _run(ctx, comp, starts, [a[:60000], a[60000:], b])
    assert info.status == 22 and info.bad_chunk == 1
    # the same first stream alone, last stretch open-ended: fine
    info, got, _ = _run(ctx, ca, sa, [a[:60000], a[60000:]])
    assert info.status == 0 and got == a


def _gz_header(name=None, extra=None, comment=None, hcrc=False):
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    h = bytes([0x1f, 0x8b, 8, flg, 1, 2, 3, 4, 0, 3])
    if extra is not None:
        h += len(extra).to_bytes(2, "little") + extra
    if name is not None:
        h += name + b"\0"
    if comment is not None:
        h += comment + b"\0"
    if hcrc:
        h += (zlib.crc32(h) & 0xffff).to_bytes(2, "little")
    return h


def test_several_members_are_decoded_in_one_go(ctx):
    """cat a.gz b.gz c.gz d.gz: a stretch that meets a member's final block skips the trailer, reads the next member's
    header (plain, with FNAME, with FEXTRA + FCOMMENT + FHCRC) and goes on; hpn_gz_members lists where the members ended
    in the text and their ISIZE.  Stretch starts: the first block of the file and two sync-flush points, one of them in the
    third member -- so one stretch holds two member ends, another none."""
    rng = np.random.default_rng(77)
    texts = [_fastq(rng, 900), _fastq(rng, 5), _fastq(rng, 2500), _fastq(rng, 700)]
    heads = [_gz_header(), _gz_header(name=b"reads_2.fq"), _gz_header(extra=b"abcdefgh", comment=b"a comment", hcrc=True), _gz_header(name=b"x" * 70)]
    comp, starts, marks = b"", [], []
    for k, (t, h) in enumerate(zip(texts, heads)):
        pieces = [t] if k != 2 else [t[:100000], t[100000:180000], t[180000:]]
        body, st = _stream(pieces, 6, zlib.Z_SYNC_FLUSH)
        at = len(comp) + len(h)
        if k == 0:
            starts.append(at)
        if k == 2:
            starts += [at + st[1], at + st[2]]
        comp += h + body + zlib.crc32(t).to_bytes(4, "little") + (len(t) & 0xffffffff).to_bytes(4, "little")
        marks.append(len(t))
    whole = b"".join(texts)
    cut1 = len(texts[0]) + len(texts[1]) + 100000
    cut2 = cut1 + 80000
    pieces = [whole[:cut1], whole[cut1:cut2], whole[cut2:]]
    info, got, _ = _run(ctx, comp, starts, pieces)
    assert info.status == 0 and got == whole and info.final_chunk == 3
    ends = np.cumsum(marks)[:-1].tolist()
    assert ctx.gz_members() == [(e, m & 0xffffffff) for e, m in zip(ends, marks[:-1])]
    # the trailer of the last member is where the last stretch stopped
    assert (starts[-1] * 8 + info.end_bit) // 8 + 8 == len(comp)
    # bytes that are not a gzip header behind a member inside a bounded stretch: status 22 as before
    bad = bytearray(comp)
    second = len(heads[0]) + len(_stream([texts[0]])[0]) + 8
    bad[second] = 0x1e
    info, got, _ = _run(ctx, bytes(bad), starts, pieces)
    assert info.status == 22 and info.bad_chunk == 0


def test_more_members_than_a_call_lists(ctx):
    """65536 member ends fit the list of one call; one more is status 23 (the tools then take the host's reader)."""
    import gzip
    one = b"@r\nACGT\n+\nIIII\n"
    for n, want_status in ((65536 + 1, 0), (65536 + 2, 23)):       # n members = n - 1 member ends inside the stretch
        blob = gzip.compress(one, 1) * n
        info, got, _ = _run(ctx, blob, [10], [one * n])
        assert info.status == want_status, (n, info.status)
        if want_status == 0:
            assert got == one * n and len(ctx.gz_members()) == n - 1


def test_damaged_files_of_several_members_never_pass_silently(ctx):
    """Bit flips anywhere in a four-member file (data, trailers, headers with names and extra fields): the call either reports
    a status, or its text is what zlib makes of the same bytes member by member (a flip inside a CRC field, a name or MTIME
    changes nothing zlib or this decoder looks at)."""
    import gzip
    rng = np.random.default_rng(99)
    texts = [_fastq(rng, 400), _fastq(rng, 3), _fastq(rng, 800), _fastq(rng, 150)]
    heads = [_gz_header(), _gz_header(name=b"b.fq"), _gz_header(extra=b"xyz", comment=b"c"), _gz_header(hcrc=True)]
    blob = b""
    for t, h in zip(texts, heads):
        blob += h + _stream([t])[0] + zlib.crc32(t).to_bytes(4, "little") + (len(t) & 0xffffffff).to_bytes(4, "little")
    whole = b"".join(texts)

    def zlib_members(raw):
        out, at = b"", 0
        try:
            while at < len(raw):
                d = zlib.decompressobj(31)
                out += d.decompress(raw[at:])
                if not d.eof:
                    return None
                at = len(raw) - len(d.unused_data)
            return out
        except zlib.error:
            return None

    assert zlib_members(blob) == whole
    first = len(heads[0])
    for trial in range(40):
        raw = bytearray(blob)
        pos = int(rng.integers(first, len(raw)))
        raw[pos] ^= 1 << int(rng.integers(0, 8))
        info, got, _ = _run(ctx, bytes(raw), [first], [whole + bytes(4096)])
        if info.status == 0:
            want = zlib_members(bytes(raw))
            # zlib also checks CRC-32 and ISIZE (the caller's job here): a flip there makes zlib fail where this call does not
            if want is not None:
                assert got == want, (trial, pos)


# ---- CRC-32 of device-resident text (hpn_crc32_dev): what gzread checks per member ------------------------------------------
def test_crc32_of_spans_equals_zlib():
    import zlib
    import torch
    import highperformancengs_amd as hp
    from highperformancengs_amd import _lib
    ctx = hp.Context(0)
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, 3_000_000, dtype=np.uint8)
    data[100_000:400_000] = 0                       # (zeros: the register must still move)
    d = torch.from_numpy(data).cuda()
    spans = [(0, 0), (0, 1), (7, 255), (1, 256), (3, 257), (13, 65535), (0, 65536), (5, 65537), (100_001, 1_048_579), (0, len(data)),
             (2_999_999, 1), (65536 * 3, 65536 * 4), (123, 4 * 65536 + 255)]
    got = ctx.crc32_dev(d, spans)
    for (o, n), c in zip(spans, got):
        assert c == zlib.crc32(data[o:o + n].tobytes()), (o, n)
    # folding: CRC(A || B) from CRC(A), CRC(B), |B|
    L = _lib.lib()
    for cut in (0, 1, 65536, 1_000_003, len(data)):
        a, b = zlib.crc32(data[:cut].tobytes()), zlib.crc32(data[cut:].tobytes())
        assert L.hpn_crc32_join(a, b, len(data) - cut) == zlib.crc32(data.tobytes())
    ctx.close()


def test_crc32_every_alignment_of_the_column_kernel():
    """Round 6's k_crc32_blocks folds 32-bit columns down rows of 4 KiB and lays a short block against the END of its 64 KiB:
    every length around a row and a block, at every offset inside a 16-byte piece, against zlib (what gzread checks behind the
    reference's gzgets, IO_stream.h:122-136)."""
    import zlib
    import torch
    import highperformancengs_amd as hp
    ctx = hp.Context(0)
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, 400_000, dtype=np.uint8)
    d = torch.from_numpy(data).cuda()
    lens = list(range(0, 70)) + [4079, 4080, 4081, 4095, 4096, 4097, 8191, 8192, 8193, 61439, 61440, 61441, 65519, 65520, 65535, 65536, 65537,
                                 69631, 69632, 131071, 131072, 131073, 200_000]
    spans = [(o, n) for n in lens for o in (0, 1, 3, 15, 16, 17, 4095)]
    got = ctx.crc32_dev(d, spans)
    bad = [(o, n) for (o, n), c in zip(spans, got) if c != zlib.crc32(data[o:o + n].tobytes())]
    assert not bad, bad[:10]
    ctx.close()


@pytest.mark.parametrize("how,n_cuts", [("lds", 60), ("global", 60), ("groups", 60), ("groups", 130), ("groups", 700)])
def test_histories_in_lds_and_in_memory(request, how, n_cuts):
    """k_gz_windows_lds (calls that have the chip to themselves), k_gz_windows and the three-step form (k_gz_win_maps / _chain /
    _apply: groups of 64 stretches composed side by side -- 60 cuts: one full group and a short one, 130: three, 700: eleven)
    resolve the same chain: forced by HPN_GZ_WINDOWS on a stream of many short stretches (placeholders ride from stretch to
    stretch; stretches shorter than 32 KiB hand the history in front of them on), with and without a history handed in."""
    from conftest import in_hooks_build
    if in_hooks_build(request, {"HPN_GZ_WINDOWS": how}):     # (the switch lives in the test-hooks library: host/knobs.hpp)
        return
    ctx = request.getfixturevalue("ctx")
    rng = np.random.default_rng(41)
    text = _fastq(rng, 12000)
    cuts = sorted(set(int(x) for x in rng.integers(1, len(text), n_cuts)) | {5, 9, 40000, 40010})
    pieces = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
    comp, starts = _stream(pieces, 6, zlib.Z_SYNC_FLUSH)
    info, got, wout = _run(ctx, comp, starts, pieces)
    assert info.status == 0 and got == text and wout == text[-32768:]
    half = len(pieces) // 2
    n1 = starts[half]
    info1, got1, w1 = _run_partial(ctx, comp, starts[:half], n1, pieces[:half])
    assert info1.status == 0 and got1 == b"".join(pieces[:half])
    info2, got2, w2 = _run(ctx, comp[n1:], [s - n1 for s in starts[half:]], pieces[half:], window=w1)
    assert info2.status == 0 and got2 == b"".join(pieces[half:]) and w2 == text[-32768:]


@pytest.mark.parametrize("seed", range(4))
def test_window_decoder_fuzz_with_unknown_history(ctx, seed):
    """The payloads of tests/test_bgzf_inflate_gpu.py::test_window_decoder_fuzz as ONE stream cut at flush points: matches
    reach in front of the stretches (history placeholders ride through the window decoder's match copies)."""
    from test_bgzf_inflate_gpu import _fuzz_payload
    rng = np.random.default_rng(9100 + seed)
    pieces = [_fuzz_payload(rng, int(k % 4), int(rng.integers(1, 90000))) for k in range(14)]
    level = int(rng.integers(1, 10))
    comp, starts = _stream(pieces, level, zlib.Z_SYNC_FLUSH if seed % 2 else zlib.Z_FULL_FLUSH,
                           [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_DEFAULT_STRATEGY][seed])
    text = b"".join(pieces)
    assert zlib.decompress(comp, -15) == text
    info, got, wout = _run(ctx, comp, starts, pieces)
    assert info.status == 0 and got == text and wout == (bytes(32768) + text)[-32768:]


def test_block_starts_found_on_the_device(ctx):
    """hpn_gz_find_starts_dev: in a stream cut at flush points the byte behind every flush is a block start; a slice that
    begins there, a little in front of it or far in front of it must report that position (or an earlier TRUE block start:
    zlib opens a new block every ~16 K symbols), and every reported start must decode from there to the stream's end."""
    rng = np.random.default_rng(77)
    text = _fastq(rng, 9000)
    cuts = sorted(int(x) for x in rng.integers(200000, len(text) - 1000, 7))
    pieces = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
    comp, starts = _stream(pieces, 6, zlib.Z_SYNC_FLUSH)
    d_comp = torch.from_numpy(np.frombuffer(comp + bytes(512), np.uint8).copy()).cuda()
    slices, want = [], []
    assert comp[starts[-1]] & 1                         # the last piece is one FINAL block: not proposed (see below)
    for s in starts[1:-1]:
        for back in (0, 3, 5 * 8 + 1, 2000 * 8):
            lo = s * 8 - back
            slices.append((lo, 60000 * 8))
            want.append(s * 8)
    slices.append((len(comp) * 8 - 80, 64))            # nothing there
    slices.append((starts[-1] * 8, 8 * (len(comp) - starts[-1] - 8)))   # a final block is not a stretch's end
    found = ctx.gz_find_starts_dev(d_comp, len(comp), slices)
    assert int(found[-1]) == (1 << 64) - 1 and int(found[-2]) == (1 << 64) - 1
    for (lo, n), w, f in zip(slices, want, found[:-2]):
        f = int(f)
        assert lo <= f <= w, (lo, w, f)
        if f != w:                                      # an earlier block start inside the slice: it must be a true one
            d = zlib.decompressobj(-15)
            # bit-unaligned start: shift the stream so that it begins on a byte (the decoder is fed from that bit on)
            bits = np.unpackbits(np.frombuffer(comp, np.uint8), bitorder="little")[f:]
            tail = np.packbits(bits, bitorder="little").tobytes()
            try:
                out = d.decompress(tail)
            except zlib.error as e:                     # a match reaching in front of the start is the only excuse
                assert "distance too far back" in str(e), e


@pytest.mark.parametrize("level", [1, 4, 6, 9])
def test_block_starts_at_any_bit_of_an_unflushed_stream(ctx, level):
    """Round 6's search (a lane owns a byte: eight bit positions per 16-byte load, the cheap header test first, the Kraft sum
    for what it leaves): on ONE deflate stream without flush points -- block starts at arbitrary bits, as in a .fastq.gz -- slices
    that begin at every bit offset of a byte and end at every bit offset must report a position inside the slice from which zlib
    decodes on (or nothing, when the slice holds no start); the same slice asked twice gives the same answer.  What gzread does
    behind the reference's gzgets (IO_stream.h:122-136) starts at the member's first bit; these are the seams of the device route."""
    rng = np.random.default_rng(100 + level)
    text = _fastq(rng, 40000, 150)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(text) + c.flush()
    d_comp = torch.from_numpy(np.frombuffer(comp + bytes(512), np.uint8).copy()).cuda()
    nbits = len(comp) * 8
    slices = []
    for k in range(48):
        lo = int(rng.integers(0, nbits - 800_000)) + k % 8
        slices.append((lo, int(rng.integers(300_000, 700_000)) + (k * 3) % 8))
    slices += [(0, 0), (nbits - 64, 64), (5, 1), (nbits - 7, 7)]
    found = ctx.gz_find_starts_dev(d_comp, len(comp), slices)
    again = ctx.gz_find_starts_dev(d_comp, len(comp), slices)
    assert [int(x) for x in found] == [int(x) for x in again]
    bits = np.unpackbits(np.frombuffer(comp, np.uint8), bitorder="little")
    hits = 0
    for (lo, n), f in zip(slices, found):
        f = int(f)
        if f == (1 << 64) - 1:
            continue
        assert lo <= f < lo + n, (lo, n, f)
        hits += 1
        tail = np.packbits(bits[f:f + 8 * 120_000], bitorder="little").tobytes()
        d = zlib.decompressobj(-15)
        try:
            out = d.decompress(tail, 60_000)
            assert len(out) >= 30_000 or d.eof
        except zlib.error as e:             # a match reaching in front of the start is the only excuse
            assert "distance too far back" in str(e), (f, e)
    assert hits >= 44, hits                 # slices of 300 K bits and more of FASTQ hold a block start (blocks are ~16 K symbols)
