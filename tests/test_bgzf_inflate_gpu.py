"""GPU parity: device-side DEFLATE decoding of BGZF blocks (hpn_bgzf_inflate_dev) against zlib.

The oracle here is zlib itself (the library the reference's bgzf.c calls): every block the GPU
inflates must equal zlib's output byte for byte, for every block type (stored, fixed, dynamic),
level and strategy zlib can emit, for overlapping matches, long codes and block-size limits;
damaged streams must be reported per block, never crash or hang."""
import struct
import zlib

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def run(ctx, streams, out_lens):
    """streams: raw DEFLATE payloads -> (list of outputs, status array)"""
    import torch
    comp = b"".join(streams) + bytes(64)
    blocks = np.zeros((len(streams), 3), np.uint64)  # in_off | in_len,out_len | out_off
    ino = outo = 0
    for i, (s, n) in enumerate(zip(streams, out_lens)):
        blocks[i, 0] = ino
        blocks[i, 1] = len(s) | (n << 32)
        blocks[i, 2] = outo
        ino += len(s)
        outo += n
    d_comp = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).cuda()
    d_blocks = torch.from_numpy(blocks.view(np.int64)).cuda()
    d_out = torch.zeros(max(outo, 1) + 64, dtype=torch.uint8, device="cuda")
    d_status = torch.full((max(len(streams), 1),), 999, dtype=torch.int32, device="cuda")
    ctx.bgzf_inflate_dev(d_comp, d_blocks, len(streams), d_out, d_status)
    ctx.sync()
    out = d_out.cpu().numpy()
    st = d_status.cpu().numpy()[:len(streams)]
    res, o = [], 0
    for n in out_lens:
        res.append(out[o:o + n].tobytes())
        o += n
    return res, st


def _payloads():
    rng = np.random.default_rng(5)
    fq = open(golden_path("fastq", "syn_var_a.fq"), "rb").read()
    p = {
        "empty": b"",
        "one": b"A",
        "text": fq[:60000],
        "text_small": fq[:777],
        "random": rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(),
        "zeros": bytes(65280),
        "period3": (b"ACG" * 22000)[:65280],
        "period70": (bytes(range(70)) * 1000)[:65000],
        "skewed": bytes(rng.choice(np.arange(256, dtype=np.uint8), 65000, p=np.r_[[0.5, 0.25, 0.125], np.full(253, 0.125 / 253)])),
        "quals": bytes(rng.integers(35, 75, 65280, dtype=np.uint8)),
        "max": rng.integers(65, 70, 65535, dtype=np.uint8).tobytes(),
        # > 256 distinct lengths of matches and far distances
        "far": (rng.integers(0, 256, 30000, dtype=np.uint8).tobytes() + fq[:2768]) * 2,
    }
    return p


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY),
                                            (6, zlib.Z_RLE), (9, zlib.Z_FILTERED)])
def test_every_block_type_level_and_strategy(ctx, level, strategy):
    p = _payloads()
    names = sorted(p)
    streams = [raw_deflate(p[k], level, strategy) for k in names]
    got, st = run(ctx, streams, [len(p[k]) for k in names])
    for k, g, s in zip(names, got, st):
        assert s == 0, (k, int(s))
        assert g == p[k], k


def test_many_small_blocks_and_memlevels(ctx):
    fq = open(golden_path("fastq", "syn_var_a.fq"), "rb").read()
    rng = np.random.default_rng(6)
    pieces, streams = [], []
    o = 0
    while o < len(fq):
        n = int(rng.integers(1, 9000))
        pieces.append(fq[o:o + n])
        streams.append(raw_deflate(pieces[-1], int(rng.integers(1, 10)), zlib.Z_DEFAULT_STRATEGY, int(rng.integers(1, 10))))
        o += n
    got, st = run(ctx, streams, [len(x) for x in pieces])
    assert not st.any()
    assert b"".join(got) == fq


@pytest.mark.parametrize("bam", ["e.bam", "rand.bam"])
def test_real_bgzf_files(ctx, bam):
    raw = open(golden_path("bam", bam), "rb").read()
    streams, lens, want = [], [], []
    o = 0
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        payload = raw[o + 12 + xlen:o + bsize - 8]
        isize = struct.unpack_from("<I", raw, o + bsize - 4)[0]
        streams.append(payload)
        lens.append(isize)
        want.append(zlib.decompress(payload, -15))
        o += bsize
    got, st = run(ctx, streams, lens)
    assert not st.any()
    assert got == want


def test_damaged_streams_are_reported_per_block(ctx):
    p = _payloads()
    good = raw_deflate(p["text"], 6)
    rng = np.random.default_rng(7)
    streams, lens = [good], [len(p["text"])]
    for k in range(40):
        b = bytearray(good)
        kind = k % 4
        if kind == 0:
            b = b[:int(rng.integers(1, len(b) - 1))]          # truncated
        elif kind == 1:
            for _ in range(3):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))  # bit flips
        elif kind == 2:
            b[0] |= 6                                            # block type 3
        else:
            b = bytearray(rng.integers(0, 256, int(rng.integers(10, 3000)), dtype=np.uint8).tobytes())  # noise
        streams.append(bytes(b))
        lens.append(len(p["text"]))
    streams.append(good)
    lens.append(len(p["text"]) - 1)   # wrong ISIZE
    streams.append(good)
    lens.append(len(p["text"]))
    got, st = run(ctx, streams, lens)
    assert st[0] == 0 and got[0] == p["text"]
    assert st[-1] == 0 and got[-1] == p["text"]   # a good block after damaged ones is unaffected
    assert st[-2] != 0
    n_bad = 0
    for i in range(1, len(streams) - 2):
        d = zlib.decompressobj(-15)
        try:
            z = d.decompress(streams[i])
            z = z if d.eof else None
        except zlib.error:
            z = None
        if st[i] == 0:  # accepted: then zlib accepts it too and yields these very bytes (a flip that keeps the stream valid)
            assert z is not None and len(z) == lens[i] and got[i] == z, i
        else:
            n_bad += 1
            assert z is None or len(z) != lens[i], i  # rejected: zlib rejects it as well (or the ISIZE check fails)
    assert n_bad >= 25


def _fuzz_payload(rng, kind, n):
    """Payloads that push the decoder's windows (inflate_core.hpp: symbols are decoded 64 bit offsets at a time) to their edges:
    code lengths from 1 to 15 bits, second-level codes, symbols that straddle a window's end, every length / distance code."""
    if kind == 0:     # geometric byte frequencies: literal codes of every length, the rare ones behind the 10-bit root table
        q = float(rng.uniform(0.3, 0.9))
        p = q ** np.arange(256)
        return bytes(rng.choice(rng.permutation(256).astype(np.uint8), n, p=p / p.sum()))
    if kind == 1:     # copies of earlier stretches at every distance and length, between literals
        out = bytearray(rng.integers(0, 256, 300, dtype=np.uint8).tobytes())
        while len(out) < n:
            if rng.random() < 0.7:
                d = int(min(len(out), rng.choice([1, 2, 3, 5, 17, 300, 1025, 4097, 16385, 32768]) + rng.integers(0, 40)))
                l = int(rng.integers(3, 259))
                s = len(out) - d
                for i in range(l):
                    out.append(out[s + i])
            else:
                out += rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes()
        return bytes(out[:n])
    if kind == 2:     # few symbols: 1- and 2-bit codes, up to 64 symbols in a window, long runs of pairs
        return bytes(rng.choice(np.frombuffer(b"AC", np.uint8), n, p=[0.9, 0.1]))
    # FASTQ-like with mutated repeats of earlier reads (lengths and distances spread over all codes)
    reads = [rng.choice(np.frombuffer(b"ACGT", np.uint8), 150).tobytes() for _ in range(40)]
    out = bytearray()
    i = 0
    while len(out) < n:
        r = bytearray(reads[int(rng.integers(0, len(reads)))])
        for _ in range(int(rng.integers(0, 4))):
            r[int(rng.integers(0, 150))] = int(rng.choice(np.frombuffer(b"ACGTN", np.uint8)))
        out += b"@r%d\n%s\n+\n%s\n" % (i, bytes(r), bytes(rng.choice(np.frombuffer(b"FFFFF:,#", np.uint8), 150)))
        i += 1
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(6))
def test_window_decoder_fuzz(ctx, seed):
    rng = np.random.default_rng(9000 + seed)
    payloads, streams = [], []
    for k in range(48):
        p = _fuzz_payload(rng, k % 4, int(rng.integers(1, 65000)))
        level = int(rng.integers(1, 10))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
        payloads.append(p)
        streams.append(raw_deflate(p, level, strategy, int(rng.integers(1, 10))))
    got, st = run(ctx, streams, [len(p) for p in payloads])
    assert not st.any(), np.flatnonzero(st)
    for k, (g, p) in enumerate(zip(got, payloads)):
        assert g == p, k
    # the same streams with the room cut short by one byte, and with one byte too much: reported, nothing written beyond
    got, st = run(ctx, streams[:8], [max(len(p) - 1, 0) for p in payloads[:8]])
    assert all(s != 0 or len(p) == 0 for s, p in zip(st, payloads[:8]))
    got, st = run(ctx, streams[:8], [len(p) + 1 for p in payloads[:8]])
    assert st.all()


@pytest.mark.parametrize("cut", [1, 2])
def test_payload_cut_inside_the_last_end_of_block_code_is_refused(ctx, cut):
    """Bytes behind in_len are read as zeros, and seven zero bits are the end-of-block code of a fixed-Huffman block: a payload
    that lost its last byte(s) decoded to the stated length with status 0 -- zlib calls such a stream unfinished.  (Found by
    scripts/soak_inflate_damaged.py in its 431st round.)"""
    import torch
    rng = np.random.default_rng(5)
    payload = bytes(rng.integers(65, 91, 3000, dtype=np.uint8))
    whole = raw_deflate(payload, 6, zlib.Z_FIXED)
    for s in (whole, whole[:-cut]):
        d = zlib.decompressobj(-15)
        z = d.decompress(s) + d.flush()
        blocks = np.array([[0, len(s) | (len(payload) << 32), 0]], np.uint64)
        d_comp = torch.from_numpy(np.frombuffer(s + bytes(64), np.uint8).copy()).cuda()
        d_blocks = torch.from_numpy(blocks.view(np.int64)).cuda()
        d_out = torch.full((len(payload) + 128,), 0xAA, dtype=torch.uint8, device="cuda")
        d_status = torch.full((1,), 999, dtype=torch.int32, device="cuda")
        ctx.bgzf_inflate_dev(d_comp, d_blocks, 1, d_out, d_status)
        ctx.sync()
        st = int(d_status.cpu()[0])
        if d.eof:
            assert st == 0 and bytes(d_out[:len(payload)].cpu().numpy()) == payload == z
        else:
            assert st != 0, "a stream zlib calls unfinished was taken"
