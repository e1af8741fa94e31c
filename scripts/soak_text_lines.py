#!/usr/bin/env python3
"""Soak of the line index (k_text_lines, round 6 form) on damaged text of several tiles: N random FASTQ texts of 0.3 - 3 MB,
one irregularity each (tests/test_fastq_text_gpu.py::_mutate and a NUL / lost / extra newline deep inside), framed from copied
chunks and in place, any chunking: every result must be the oracle's gzgets loop's, or the text refused.

    python3 scripts/soak_text_lines.py [N=200]  -> one JSON line"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import highperformancengs_amd as hp  # noqa: E402
import orc  # noqa: E402
import test_fastq_text_gpu as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = hp.Context(0)
td = tempfile.mkdtemp(prefix="soak_text_")
refused = exact = clean_taken = 0
for i in range(N):
    rng = np.random.default_rng(90_000 + i)
    text = b"".join(T._random_fastq(rng, int(rng.integers(1500, 15000)), 20, int(rng.integers(60, 300))))
    kind = i % 5
    if kind == 0:
        pass
    elif kind == 1:
        b = bytearray(text)
        b[int(rng.integers(len(b) // 8, len(b)))] = 0
        text = bytes(b)
    else:
        text = T._mutate(rng, text)
    p = os.path.join(td, "s.fq")
    open(p, "wb").write(text)
    rc, want = orc.count_stream(p)
    for size in (None, int(rng.integers(100_000, 900_000))):
        for fn in (lambda: T._count(ctx, text, size, tail_call=bool(size)), lambda: T._count_inplace(ctx, text, size)):
            res, flags, n = fn()
            if res is None:
                refused += 1
                assert kind != 0, "undamaged text refused"
            else:
                assert rc == 0, (i, kind)
                T._assert_counts(res, want)
                exact += 1
                clean_taken += kind == 0
            if kind == 1:
                assert res is None, "a NUL byte went unnoticed"
os.remove(p)
os.rmdir(td)
ctx.close()
print(json.dumps({"texts": N, "framings": 4 * N, "refused": refused, "exact": exact, "undamaged_taken": clean_taken}))
