#!/usr/bin/env python3
"""k_gz_sym_inflate / k_gz_windows / k_gz_translate at chip scale: 6,144 stretches of a deflate stream in one call (one per decoder wave of the chip).

The stretches are cut at Z_SYNC_FLUSH points (byte-aligned block starts, later stretches refer back into earlier ones, so
the symbolic output does hold history placeholders); the table repeats the stream's ~50 stretches round-robin to fill the
chip (every entry has its own output region; the first pass over the stream is checked against the text)."""
import json
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import highperformancengs_amd as hp  # noqa: E402

n_stretch = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
rng = np.random.default_rng(1)
n, L = 300_000, 100
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, L))
qual = rng.integers(35, 74, (n, L), dtype=np.uint8)
text = b"".join(b"@read%d/1\n%s\n+\n%s\n" % (i, seq[i].tobytes(), qual[i].tobytes()) for i in range(n))
step = 1_250_000
pieces = [text[i:i + step] for i in range(0, len(text), step)]
c = zlib.compressobj(6, zlib.DEFLATED, -15)
comp, starts = b"", []
for k, p in enumerate(pieces):
    starts.append(len(comp))
    comp += c.compress(p) + (c.flush(zlib.Z_SYNC_FLUSH) if k + 1 < len(pieces) else c.flush())
m = len(pieces)
CH = np.dtype([("in_off", "<u8"), ("end_bit", "<u8"), ("in_len", "<u4"), ("start_bit", "<u4")])
tab = np.zeros(n_stretch, CH)
lens = []
for j in range(n_stretch):
    k = j % m
    tab[j]["in_off"] = starts[k]
    tab[j]["in_len"] = len(comp) - starts[k]
    tab[j]["end_bit"] = (starts[k + 1] - starts[k]) * 8 if k + 1 < m else (1 << 64) - 1
    lens.append(len(pieces[k]))
total = sum(lens)
cap = (step + 4096 + 7) // 8 * 8
ctx = hp.Context(0)
d_comp = torch.from_numpy(np.frombuffer(comp + bytes(256), np.uint8).copy()).cuda()
d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
d_text = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
ms = []
for rep in range(4):
    t0 = torch.cuda.Event(enable_timing=True)
    info = ctx.gz_inflate_dev(d_comp, d_tab, n_stretch, cap, d_text, total + 64)
    assert info.status == 0 and info.n_bytes == total, (info.status, info.bad_chunk, info.n_bytes, total)
    if rep:
        ms.append(ctx.last_kernel_ms(5))
got = d_text[:len(text)].cpu().numpy().tobytes() if n_stretch >= m else b""
ok = (got == text) if n_stretch >= m else None
k_ms = float(np.median(ms))
comp_bytes = sum((starts[(j % m) + 1] if (j % m) + 1 < m else len(comp)) - starts[j % m] for j in range(n_stretch))
print(json.dumps({"kernel": "k_gz_sym_inflate", "stretches": n_stretch, "text_bytes": total, "compressed_bytes": comp_bytes,
                  "kernel_ms": round(k_ms, 2), "text_GBps": round(total / k_ms / 1e6, 2), "compressed_GBps": round(comp_bytes / k_ms / 1e6, 2),
                  "symbol_bytes_written": 2 * total, "first_pass_equals_text": ok}))
ctx.close()
