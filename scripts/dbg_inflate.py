import sys, zlib, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import highperformancengs_amd as hp
from test_bgzf_inflate_gpu import run, raw_deflate
ctx = hp.Context(0)
rng = np.random.default_rng(10)
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (200, 100))
qual = rng.integers(35, 74, (200, 100), dtype=np.uint8)
text = b"".join(b"@read%d/1\n%s\n+\n%s\n" % (i, seq[i].tobytes(), qual[i].tobytes()) for i in range(200))
for level in (1, 6):
    s = raw_deflate(text, level)
    got, st = run(ctx, [s], [len(text)])
    g = np.frombuffer(got[0], np.uint8); w = np.frombuffer(text, np.uint8)
    bad = np.flatnonzero(g != w)
    print("level", level, "status", st, "mismatches", len(bad), bad[:40])
    for b in bad[:6]:
        print(b, bytes(w[max(0,b-12):b+6]), bytes(g[max(0,b-12):b+6]))
