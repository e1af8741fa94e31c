#!/usr/bin/env python3
"""k_gz_find_starts and k_crc32_blocks at the scale of one batch of the gzip route (round 6).

Input: gzip members as BASELINE configs[1] names them (zlib level 1, 1e5 reads x 150 bp per member), laid back to back on the
device until one batch is filled (6,144 slices of 1.5 MiB by default, as host/gz_gpu.hpp cuts a 186 GB file), one slice per
decoder slot.  Every reported start is checked: zlib must decode from that bit to the member's end.  Then CRC-32 of the text of
the members against zlib's.   python scripts/bench_gz_find.py [slices] [slice_bytes]
A library built with -DHPN_FIND_DIAG (HPN_LIB=...) prints where the search's clocks went."""
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import highperformancengs_amd as hp  # noqa: E402
import bench_extra  # noqa: E402

n_slices = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
slice_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 3 << 19
K, per, L = 12, 100_000, 150
ctx = hp.Context(0)
texts = [bench_extra._fastq_text(ctx, per, L, 100 + k).tobytes() for k in range(K)]
with ThreadPoolExecutor(12) as ex:
    members = list(ex.map(bench_extra._gzip_one, texts))
cycle = b"".join(members)
need = n_slices * slice_bytes + 4 * slice_bytes
reps = (need + len(cycle) - 1) // len(cycle)
d_cycle = torch.from_numpy(np.frombuffer(cycle, np.uint8).copy()).cuda()
d_comp = torch.cat([d_cycle.repeat(reps), torch.zeros(4096, dtype=torch.uint8, device="cuda")])
comp_len = reps * len(cycle)
slices = [((k + 1) * slice_bytes * 8, slice_bytes * 8) for k in range(n_slices)]
res = {"slices": n_slices, "slice_bytes": slice_bytes, "compressed_GB": round(n_slices * slice_bytes / 1e9, 2)}
times = []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    found = ctx.gz_find_starts_dev(d_comp, comp_len, slices)
    times.append((time.perf_counter() - t0) * 1e3)
res["find_ms"] = [round(t, 2) for t in times]
found = np.asarray(found, dtype=np.uint64)
res["found"] = int((found != np.uint64(2**64 - 1)).sum())
dist = (found - np.array([s[0] for s in slices], dtype=np.uint64)).astype(np.float64) / 8
res["bytes_scanned_mean_max"] = [round(float(dist.mean())), round(float(dist.max()))]
# every 97th start: zlib decodes from that bit on to the member's end
bits = np.unpackbits(np.frombuffer(cycle, np.uint8), bitorder="little")
bad = 0
for f in found[::97]:
    at = int(f) % (len(cycle) * 8)
    tail = np.packbits(bits[at:at + 8 * 400_000], bitorder="little").tobytes()
    d = zlib.decompressobj(-15)
    try:
        out = d.decompress(tail, 200_000)
        bad += len(out) < 100_000
    except zlib.error as e:
        bad += "distance too far back" not in str(e)
res["checked_by_zlib"] = {"starts": len(found[::97]), "bad": bad}
# ---- CRC-32 ----
text = b"".join(texts)
want = [zlib.crc32(t) for t in texts]
d_text = torch.from_numpy(np.frombuffer(text, np.uint8).copy()).cuda()
treps = max(1, int(3e9 // len(text)))
d_big = d_text.repeat(treps)
spans, at = [], 0
for r in range(treps):
    for t in texts:
        spans.append((at, len(t)))
        at += len(t)
ctimes = []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = ctx.crc32_dev(d_big, spans)
    ctimes.append((time.perf_counter() - t0) * 1e3)
res["crc_ms"] = [round(t, 2) for t in ctimes]
res["crc_GB"] = round(at / 1e9, 2)
res["crc_GBps_best"] = round(at / min(ctimes) / 1e6, 1)
res["crc_equal_zlib"] = bool(all(int(g) == want[i % K] for i, g in enumerate(got)))
# ragged spans: odd offsets and lengths
rng = np.random.default_rng(3)
rs = [(int(o), int(n)) for o, n in zip(rng.integers(0, 5_000_000, 40), rng.integers(0, 300_000, 40))] + [(7, 0), (1, 1), (65535, 65536), (3, 65537)]
got = ctx.crc32_dev(d_text, rs)
res["crc_ragged_equal_zlib"] = bool(all(int(g) == zlib.crc32(text[o:o + n]) for (o, n), g in zip(rs, got)))
print(json.dumps(res))
ctx.close()
