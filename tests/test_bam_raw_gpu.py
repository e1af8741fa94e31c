"""GPU parity through the C ABI: BAM files taken as they are -- BGZF blocks inflated on the device
(hpn_bgzf_inflate_dev), records indexed in place (hpn_bam_raw_index_dev), depth / window kernels
run on the raw records -- against the oracle's dense model fed by the Python BAM decoder."""
import struct
import zlib

import numpy as np
import pytest

import orc
from conftest import golden_path
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


def _blocks(raw):
    """BGZF member chain -> list of (payload_off, payload_len, isize)"""
    out, o = [], 0
    while o < len(raw):
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        out.append((o + 12 + xlen, bsize - xlen - 20, struct.unpack_from("<I", raw, o + bsize - 4)[0]))
        o += bsize
    return out


def _header_len(text):
    l_text = struct.unpack_from("<i", text, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", text, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", text, p)[0]
    return p


def _to_device(ctx, raw, skip_header=True):
    """-> (d_raw, info, keepalive) with the blocks from the one holding the first record on"""
    import torch
    blks = _blocks(raw)
    text, hl = b"", None
    for a, n, _ in blks:                                    # as many blocks as the header takes (tiny blocks: several)
        text += zlib.decompress(raw[a:a + n], -15)
        try:
            hl = _header_len(text)
            break
        except struct.error:
            continue
    first, acc = 0, 0
    while acc + blks[first][2] <= hl and first < len(blks) - 1:  # the block the header ends in (or the next one)
        acc += blks[first][2]
        first += 1
    blks = blks[first:]
    table = np.zeros((len(blks), 3), np.uint64)
    outo = 0
    for i, (a, n, isz) in enumerate(blks):
        table[i] = (a, n | (isz << 32), outo)
        outo += isz
    d_comp = torch.from_numpy(np.frombuffer(raw + bytes(64), np.uint8).copy()).cuda()
    d_blocks = torch.from_numpy(table.view(np.int64)).cuda()
    d_out = torch.zeros(outo + 64, dtype=torch.uint8, device="cuda")
    d_status = torch.zeros(len(blks), dtype=torch.int32, device="cuda")
    ctx.bgzf_inflate_dev(d_comp, d_blocks, len(blks), d_out, d_status)
    info = ctx.bam_raw_index_dev(d_out, d_blocks, len(blks), hl - acc, d_status)
    return d_out, info, (d_comp, d_blocks, d_status)


@pytest.mark.parametrize("bam", ["e.bam", "rand.bam"])
def test_index_depth_and_window_on_raw_records(ctx, bam):
    raw = open(golden_path("bam", bam), "rb").read()
    soa = bamio.read_bam_records(golden_path("bam", bam))
    d_raw, info, keep = _to_device(ctx, raw)
    assert info.flags == 0 and info.n_records == len(soa.tid)
    assert info.tid_min == int(soa.tid.min()) and info.tid_max == int(soa.tid.max())
    for W in (100, 20000):
        for tid, (name, tlen) in enumerate(soa.refs):
            for mask in (0x704, 0x4):
                runs, win = ctx.depth_target_raw(d_raw, tid, tlen, W, mask)
                rc, wruns, wbins = orc.depth_target(soa, tid, W, mask)
                assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins), (name, W, mask)
        off = orc.window_offsets(soa.refs, W)
        got = ctx.window_counts_raw(d_raw, off, W)
        want = orc.window_counts(soa, W)  # rc, off, bins, gc, len, touched, n_count
        assert want[0] == 0
        for g, w in zip(got[:4], want[2:6]):
            assert np.array_equal(np.asarray(g), np.asarray(w))
        assert got[4] == want[6]


def _packed(raw, block):
    """the same uncompressed stream cut into BGZF blocks of `block` bytes: records now run across block ends (htsjdk's way)"""
    data = b"".join(zlib.decompress(raw[a:a + n], -15) for a, n, _ in _blocks(raw))
    packed = b""
    for i in range(0, len(data), block):
        piece = data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(piece) + co.flush()
        packed += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (len(comp) + 25).to_bytes(2, "little") + comp +
                   (zlib.crc32(piece) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little"))
    return packed, data


@pytest.mark.parametrize("block", [97, 1000, 20000, 65536])
def test_records_straddling_blocks_are_decoded(ctx, block):
    """htsjdk-style BAM (records packed across BGZF blocks; block = 97: every record covers several blocks, most blocks hold
    no record start): every block's first record is found and the chain proven on the device (k_raw_starts / k_raw_scan), the
    depth and window kernels then see the same records as with samtools' record-aligned blocks."""
    raw = open(golden_path("bam", "rand.bam"), "rb").read()
    soa = bamio.read_bam_records(golden_path("bam", "rand.bam"))
    packed, _ = _packed(raw, block)
    d_raw, info, keep = _to_device(ctx, packed)
    assert info.flags & 3 == 0 and info.flags & 4 and info.tail_bytes == 0
    assert info.n_records == len(soa.tid) and info.tid_min == int(soa.tid.min()) and info.tid_max == int(soa.tid.max())
    W = 100
    for tid, (name, tlen) in enumerate(soa.refs):
        runs, win = ctx.depth_target_raw(d_raw, tid, tlen, W, 0x704)
        rc, wruns, wbins = orc.depth_target(soa, tid, W, 0x704)
        assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins), name
    got = ctx.window_counts_raw(d_raw, orc.window_offsets(soa.refs, W), W)
    want = orc.window_counts(soa, W)
    for g, w in zip(got[:4], want[2:6]):
        assert np.array_equal(np.asarray(g), np.asarray(w))
    assert got[4] == want[6]


@pytest.mark.parametrize("block,cut", [(20000, 2), (1000, 37), (97, 400), (97, 401), (97, 402), (97, 403)])
def test_unfinished_record_at_the_end_of_a_call_is_reported_and_carried(ctx, block, cut):
    """A call that ends inside a record indexes the whole records and reports the bytes of the unfinished one
    (hpn_raw_info.tail_bytes); with those bytes in front of the next call's stream (blocks moved up, first_off = 0) the two
    calls together index every record of the file -- what host/bam_gpu.hpp does from launch to launch."""
    import torch
    raw = open(golden_path("bam", "rand.bam"), "rb").read()
    soa = bamio.read_bam_records(golden_path("bam", "rand.bam"))
    packed, data = _packed(raw, block)
    blks = _blocks(packed)
    hl = _header_len(data)
    first = hl // block                                     # the block the first record starts in
    rec_at = [hl]
    while rec_at[-1] < len(data):
        rec_at.append(rec_at[-1] + 4 + struct.unpack_from("<i", data, rec_at[-1])[0])
    d_comp = torch.from_numpy(np.frombuffer(packed + bytes(64), np.uint8).copy()).cuda()

    def call(lo, hi, front, first_off):
        """blocks [lo, hi) behind `front` carried bytes -> (info, inflated stream as bytes)"""
        table = np.zeros((hi - lo, 3), np.uint64)
        outo = len(front)
        for i, (a, n, isz) in enumerate(blks[lo:hi]):
            table[i] = (a, n | (isz << 32), outo)
            outo += isz
        d_blocks = torch.from_numpy(table.view(np.int64)).cuda()
        d_out = torch.zeros(outo + 64, dtype=torch.uint8, device="cuda")
        if front:
            d_out[:len(front)] = torch.from_numpy(np.frombuffer(front, np.uint8).copy()).cuda()
        d_status = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
        ctx.bgzf_inflate_dev(d_comp, d_blocks, hi - lo, d_out, d_status)
        info = ctx.bam_raw_index_dev(d_out, d_blocks, hi - lo, first_off, d_status)
        return info, bytes(d_out[:outo].cpu().numpy())

    mid = first + cut
    end_a = min(mid * block, len(data))
    n_a = sum(1 for k in range(len(rec_at) - 1) if rec_at[k + 1] <= end_a)          # records whole in the first call
    tail = end_a - rec_at[n_a]
    info, stream = call(first, mid, b"", hl - first * block)
    assert info.flags & 3 == 0 and info.n_records == n_a and info.tail_bytes == tail, (info.flags, info.n_records, info.tail_bytes, tail)
    front = stream[len(stream) - tail:] if tail else b""
    info2, _ = call(mid, len(blks), front, 0)
    assert info2.flags & 3 == 0 and info2.tail_bytes == 0 and info2.n_records == len(soa.tid) - n_a
    assert min(info.tid_min, info2.tid_min) == int(soa.tid.min()) and max(info.tid_max, info2.tid_max) == int(soa.tid.max())


def _bgzf_pack(data, block):
    out = b""
    for i in range(0, len(data), block):
        piece = data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(piece) + co.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (len(comp) + 25).to_bytes(2, "little") + comp +
                (zlib.crc32(piece) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little"))
    return out


@pytest.mark.parametrize("block", [1000, 20000])
def test_records_nobody_guesses_are_walked_by_the_proof(ctx, block):
    """A guess asks for a printable read name (what makes four-in-a-row rare enough to trust); bam_read1 does not, and neither does
    the proof: with every name made of control characters and bytes above 127 no block of a packed file gets a guess, the chain is
    walked block by block from the call's first record, and the result is the same."""
    raw = open(golden_path("bam", "rand.bam"), "rb").read()
    soa = bamio.read_bam_records(golden_path("bam", "rand.bam"))
    data = bytearray(b"".join(zlib.decompress(raw[a:a + n], -15) for a, n, _ in _blocks(raw)))
    at = _header_len(bytes(data))
    k = 0
    while at < len(data):
        bs = struct.unpack_from("<i", data, at)[0]
        l_name = data[at + 12]
        for j in range(l_name - 1):
            data[at + 36 + j] = (1, 7, 0xe9, 0xff, 31, 127)[(k + j) % 6]
        at += 4 + bs
        k += 1
    d_raw, info, keep = _to_device(ctx, _bgzf_pack(bytes(data), block))
    assert info.flags & 3 == 0 and info.tail_bytes == 0 and info.n_records == len(soa.tid)
    W = 100
    for tid, (name, tlen) in enumerate(soa.refs):
        runs, win = ctx.depth_target_raw(d_raw, tid, tlen, W, 0x704)
        rc, wruns, wbins = orc.depth_target(soa, tid, W, 0x704)
        assert rc == 0 and np.array_equal(runs, wruns) and np.array_equal(win.astype(np.float64), wbins), name


def test_a_start_that_is_not_one_is_refuted(ctx):
    """The found starts are guesses until the chain proves them: a block whose bytes hold a well-formed chain of records at
    the WRONG place (here: a block of the file pasted in the middle of a long record's qualities would do the same) must
    flag the call, not shift the records.  Built by dropping one block of a packed file: the chain of the block before no
    longer arrives at the next found start."""
    raw = open(golden_path("bam", "rand.bam"), "rb").read()
    packed, _ = _packed(raw, 1000)
    blks, o, cut = _blocks(packed), 0, []
    for k in range(len(blks)):
        bsize = struct.unpack_from("<H", packed, o + 16)[0] + 1
        cut.append((o, bsize))
        o += bsize
    k = len(blks) // 2
    dropped = packed[:cut[k][0]] + packed[cut[k][0] + cut[k][1]:]
    _, info, _ = _to_device(ctx, dropped)
    assert info.flags & 1 and info.flags & 2 == 0


def test_damaged_block_is_flagged(ctx):
    raw = bytearray(open(golden_path("bam", "rand.bam"), "rb").read())
    a, n, _ = _blocks(bytes(raw))[-3]  # a data block behind the ones the helper reads the header from
    for k in range(a + 10, a + 40):
        raw[k] ^= 0xa5
    _, info, _ = _to_device(ctx, bytes(raw))
    assert info.flags & 2


def _reblock(raw, patch):
    """Same BGZF block boundaries, payload of every block passed through patch(block_index, bytearray)."""
    out = b""
    for i, (a, n, _) in enumerate(_blocks(raw)):
        piece = bytearray(zlib.decompress(raw[a:a + n], -15))
        patch(i, piece)
        piece = bytes(piece)
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(piece) + co.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (len(comp) + 25).to_bytes(2, "little") + comp +
                (zlib.crc32(piece) & 0xffffffff).to_bytes(4, "little") + len(piece).to_bytes(4, "little"))
    return out


@pytest.mark.parametrize("field", ["l_seq_huge", "l_seq_negative", "n_cigar", "l_name"])
def test_record_that_lies_about_its_fields_is_flagged(ctx, field, tmp_path):
    """A record whose l_read_name / n_cigar_op / l_seq do not fit its block_size must never reach the kernels
    that read name, CIGAR and sequence in place (they would read out of bounds): k_raw_count flags the file."""
    import os
    import subprocess
    raw = open(golden_path("bam", "rand.bam"), "rb").read()

    def patch(i, piece):
        if i != 1:
            return
        at = 0
        for _ in range(5):                               # the sixth record of the second block
            at += 4 + struct.unpack_from("<i", piece, at)[0]
        if field == "l_seq_huge":
            struct.pack_into("<i", piece, at + 20, 0x7fffff00)
        elif field == "l_seq_negative":
            struct.pack_into("<i", piece, at + 20, -5)
        elif field == "n_cigar":
            struct.pack_into("<H", piece, at + 16, 65535)
        else:
            piece[at + 12] = 255
    bad = _reblock(raw, patch)
    _, info, _ = _to_device(ctx, bad)
    assert info.flags & 1
    # through the tools: the GPU ingest is abandoned, the host reader stops at the record with a message
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "bad.bam").write_bytes(bad)
    (tmp_path / "bad.bam.bai").write_bytes(open(golden_path("bam", "rand.bam.bai"), "rb").read())
    for tool, args in (("bam2depth", ["-o", "d", "bad.bam"]), ("bam_sliding_count", ["-o", "s", "bad.bam"])):
        p = subprocess.run([os.path.join(root, "highperformancengs_amd", "bin", tool)] + args, cwd=tmp_path,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1"})
        assert p.returncode == 0, p.stderr.decode()
        assert b"corrupt BAM record" in p.stderr and b"host ingest" in p.stderr
