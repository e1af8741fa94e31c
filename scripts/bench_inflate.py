#!/usr/bin/env python3
"""Device-side BGZF inflate rate on a BAM / bgzip file already in HBM (compressed in, inflated out)."""
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

path = sys.argv[1]
limit = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 62
raw = np.fromfile(path, np.uint8, count=min(os.path.getsize(path), limit))
t0 = time.perf_counter()
blocks = []
o = outo = 0
n = len(raw)
mv = memoryview(raw)
while o + 18 <= n:
    xlen = mv[o + 10] | mv[o + 11] << 8
    bsize = (mv[o + 16] | mv[o + 17] << 8) + 1
    if o + bsize > n:
        break
    isize = struct.unpack_from("<I", mv, o + bsize - 4)[0]
    blocks.append((o + 12 + xlen, (bsize - 12 - xlen - 8) | isize << 32, outo))
    outo += isize
    o += bsize
t_walk = time.perf_counter() - t0
blocks = np.array(blocks, np.uint64)
ctx = hp.Context(0)
d_comp = torch.from_numpy(np.concatenate([raw[:o], np.zeros(64, np.uint8)])).cuda()
d_blocks = torch.from_numpy(blocks.view(np.int64)).cuda()
d_out = torch.empty(outo + 64, dtype=torch.uint8, device="cuda")
d_status = torch.zeros(len(blocks), dtype=torch.int32, device="cuda")
for rep in range(3):
    ctx.bgzf_inflate_dev(d_comp, d_blocks, len(blocks), d_out, d_status)
    ctx.sync()
    ms = ctx.last_kernel_ms(5)
bad = int((d_status != 0).sum().item())
print(json.dumps({"file": os.path.basename(path), "blocks": len(blocks), "compressed_bytes": int(o), "inflated_bytes": int(outo),
                  "kernel_ms": round(ms, 3), "inflated_GBps": round(outo / ms / 1e6, 2), "compressed_GBps": round(o / ms / 1e6, 2),
                  "bad_blocks": bad, "python_header_walk_s": round(t_walk, 2)}))
if len(sys.argv) > 3:  # verify against zlib on a sample of blocks
    import zlib
    out = d_out.cpu().numpy()
    rng = np.random.default_rng(1)
    for i in rng.integers(0, len(blocks), 200):
        a, w, oo = (int(x) for x in blocks[i])
        want = zlib.decompress(raw[a:a + (w & 0xffffffff)].tobytes(), -15)
        assert out[oo:oo + (w >> 32)].tobytes() == want, i
    print("sample of 200 blocks identical to zlib")
