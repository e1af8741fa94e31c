"""Soak of the window decoder on DAMAGED streams (round 5: the decoder's loop, queue and emit are new): valid raw-deflate streams of
the fuzz payloads (tests/test_bgzf_inflate_gpu.py::_fuzz_payload) with 1 - 3 random bits flipped, or cut short, thousands of them.
Properties: every launch returns (no hang: a symbol emits at least one unit or ends the window; output is bounded by out_len), a
block whose status is 0 holds exactly what zlib makes of the same damaged bits, and a block zlib decodes to the stated length is
decoded the same (status 0) -- nothing is written beyond a block's out_len (the bytes behind every block stay 0xAA).
    timeout 600 python scripts/soak_inflate_damaged.py [rounds]"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import highperformancengs_amd as hp  # noqa: E402
from test_bgzf_inflate_gpu import _fuzz_payload, raw_deflate  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = hp.Context(0)
rng = np.random.default_rng(20251004)
GAP = 64
n_streams = n_ok = n_bad = n_same_as_zlib = 0
for r in range(rounds):
    payloads = [_fuzz_payload(rng, k % 4, int(rng.integers(1, 65000))) for k in range(64)]
    streams = []
    for p in payloads:
        s = bytearray(raw_deflate(p, int(rng.integers(1, 10)), [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]))
        how = int(rng.integers(0, 4))
        if how < 3:
            for _ in range(how + 1):
                b = int(rng.integers(0, len(s) * 8))
                s[b >> 3] ^= 1 << (b & 7)
        else:
            s = s[:max(1, int(rng.integers(1, len(s) + 1)))]
        streams.append(bytes(s))
    comp = b"".join(streams) + bytes(64)
    blocks = np.zeros((len(streams), 3), np.uint64)
    ino = outo = 0
    for i, (s, p) in enumerate(zip(streams, payloads)):
        blocks[i] = (ino, len(s) | (len(p) << 32), outo)
        ino += len(s)
        outo += len(p) + GAP
    d_comp = torch.from_numpy(np.frombuffer(comp, np.uint8).copy()).cuda()
    d_blocks = torch.from_numpy(blocks.view(np.int64)).cuda()
    d_out = torch.full((outo + 64,), 0xAA, dtype=torch.uint8, device="cuda")
    d_status = torch.full((len(streams),), 999, dtype=torch.int32, device="cuda")
    ctx.bgzf_inflate_dev(d_comp, d_blocks, len(streams), d_out, d_status)
    ctx.sync()
    out, st = d_out.cpu().numpy(), d_status.cpu().numpy()
    for i, (s, p) in enumerate(zip(streams, payloads)):
        oo, n = int(blocks[i, 2]), len(p)
        assert st[i] != 999, (r, i)
        assert (out[oo + n:oo + n + GAP] == 0xAA).all(), ("wrote beyond out_len", r, i, int(st[i]))
        try:
            d = zlib.decompressobj(-15)
            z = d.decompress(s) + d.flush()
            z_ok = d.eof and len(z) == n
        except zlib.error:
            z, z_ok = b"", False
        if st[i] == 0:
            assert z_ok and out[oo:oo + n].tobytes() == z, ("status 0 but zlib disagrees", r, i)
            n_ok += 1
        else:
            n_bad += 1
            # zlib accepts and makes the stated length: so must the decoder (bytes behind the final block are not its business)
            assert not z_ok, ("zlib decodes it to the stated length, the decoder refused", r, i, int(st[i]))
        n_same_as_zlib += 1
    n_streams += len(streams)
    if r % 10 == 9:
        print(f"round {r + 1}: {n_streams} damaged streams, {n_ok} still decode (= zlib), {n_bad} refused", flush=True)
print(f"damaged streams: {n_streams}; decoded like zlib: {n_ok}; refused like zlib: {n_bad}; none hung, none wrote beyond its block")
