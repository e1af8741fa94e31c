"""Soak of the window decoder beyond the test suite: every block of a 2 GB BAM (/tmp/abw/a.bam, scripts/ab_inflate.sh makes it) against
zlib, and 1,920 more fuzz streams (tests/test_bgzf_inflate_gpu.py::_fuzz_payload).  Round 3: 59,136 blocks, none differing; all streams equal."""
import sys, os, zlib, struct, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import highperformancengs_amd as hp
from test_bgzf_inflate_gpu import _fuzz_payload, raw_deflate, run
ctx = hp.Context(0)
# 1. every block of a 2 GB BAM against zlib
path = "/tmp/abw/a.bam"
raw = np.fromfile(path, np.uint8)
blocks = []; o = outo = 0; n = len(raw); mv = memoryview(raw)
while o + 18 <= n:
    xlen = mv[o + 10] | mv[o + 11] << 8
    bsize = (mv[o + 16] | mv[o + 17] << 8) + 1
    isize = struct.unpack_from("<I", mv, o + bsize - 4)[0]
    blocks.append((o + 12 + xlen, (bsize - 12 - xlen - 8) | isize << 32, outo)); outo += isize; o += bsize
blocks = np.array(blocks, np.uint64)
d_comp = torch.from_numpy(np.concatenate([raw[:o], np.zeros(64, np.uint8)])).cuda()
d_blocks = torch.from_numpy(blocks.view(np.int64)).cuda()
d_out = torch.empty(outo + 64, dtype=torch.uint8, device="cuda")
d_status = torch.zeros(len(blocks), dtype=torch.int32, device="cuda")
ctx.bgzf_inflate_dev(d_comp, d_blocks, len(blocks), d_out, d_status); ctx.sync()
assert int((d_status != 0).sum().item()) == 0
out = d_out.cpu().numpy()
bad = 0
for i in range(len(blocks)):
    a, w, oo = (int(x) for x in blocks[i])
    if out[oo:oo + (w >> 32)].tobytes() != zlib.decompress(raw[a:a + (w & 0xffffffff)].tobytes(), -15): bad += 1
print("BAM blocks", len(blocks), "differing from zlib:", bad, flush=True)
# 2. more fuzz seeds
tot = 0
for seed in range(100, 140):
    rng = np.random.default_rng(seed)
    payloads, streams = [], []
    for k in range(48):
        p = _fuzz_payload(rng, k % 4, int(rng.integers(1, 65000)))
        payloads.append(p)
        streams.append(raw_deflate(p, int(rng.integers(1, 10)), [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))], int(rng.integers(1, 10))))
    got, st = run(ctx, streams, [len(p) for p in payloads])
    assert not st.any(), (seed, np.flatnonzero(st))
    for k, (g, p) in enumerate(zip(got, payloads)):
        assert g == p, (seed, k)
    tot += len(payloads)
print("fuzz streams ok:", tot)
