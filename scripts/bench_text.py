#!/usr/bin/env python3
"""Device-side FASTQ framing (hpn_fastq_text_count / _trim) on text already in HBM: the rate the
raw-text front end sustains without the PCIe copy in front of it.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations (k_text_lines, k_text_records,
k_text_gather, k_text_trim)."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))  # only to write the synthetic input file
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
reads, rl = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000, 150
chunk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 128 << 20
reps = 5
td = tempfile.mkdtemp(prefix="hpn_text_")
path = os.path.join(td, "s.fq")
L.orc_synth_write_fastq(path.encode(), 5, 0, reads, rl, rl, 0)
text = torch.from_numpy(np.fromfile(path, np.uint8)).cuda()
os.unlink(path)
os.rmdir(td)
ctx = hp.Context(0)
parts = [text[i:i + chunk] for i in range(0, text.numel(), chunk)]
out = torch.empty(chunk + 8192, dtype=torch.uint8, device="cuda")


def count():
    ctx.text_begin()
    n = 0
    for i, p in enumerate(parts):
        info = ctx.text_count(p, last=(i == len(parts) - 1))
        assert info.irregular == 0
        n += info.n_records
    r = ctx.fastq_tally_fetch()
    assert n == reads and r.total == reads * rl
    return n


def trim():
    import ctypes
    from highperformancengs_amd import _lib
    ctx.text_begin()
    n = 0
    for i, p in enumerate(parts):
        info = _lib.TextInfo()
        rc = ctx.L.hpn_fastq_text_trim(ctx.h, ctypes.c_void_p(p.data_ptr()), p.numel(), int(i == len(parts) - 1), 5, 140,
                                       ctypes.c_void_p(out.data_ptr()), out.numel(), ctypes.byref(info))
        assert rc == 0 and info.irregular == 0
        n += info.n_records
    assert n == reads
    return n


for name, fn in (("text_count", count), ("text_trim", trim)):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"path": name, "text_bytes": text.numel(), "chunk_bytes": chunk, "ms": round(dt * 1e3, 3),
                      "text_GBps": round(text.numel() / dt / 1e9, 1), "Gbases_per_s": round(reads * rl / dt / 1e9, 1)}), flush=True)
