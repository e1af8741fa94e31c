import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import highperformancengs_amd as hp
import c4, orc, tempfile
td = tempfile.mkdtemp()
tg = [("chr1", 3_000_000, 400_000), ("chrM", 16569, 3300), ("chrEmpty", 70_000, 0), ("chr9", 1_200_000, 100_000)]
bam, prefix = c4.synth(td, "s.bam", tg, 4)
soa = c4.Soa(prefix, len(tg))
v = c4.whole_view(soa, tg)
W = 20000
rc, off, wb, wg, wl, wt, wn = orc.window_counts(v, W)
ctx = hp.Context(0)
bins, gc, ln, touched, nc = ctx.window_counts(v, off, W)
print("bins equal", np.array_equal(bins, wb), "len equal", np.array_equal(ln, wl), "gc equal", np.array_equal(gc, wg), nc, wn)
bad = np.nonzero(gc != wg)[0]
print("windows with wrong gc:", len(bad), bad[:20], (gc[bad[:20]].astype(np.int64) - wg[bad[:20]].astype(np.int64)))
# which records fall there
lo = np.searchsorted(soa.tid, np.arange(len(tg) + 1))
for w in bad[:6]:
    t = int(np.searchsorted(off, w, side="right") - 1)
    k = int(w - off[t])
    idx = np.nonzero((soa.tid == t) & (soa.pos // W == k))[0]
    print("window", w, "target", t, "k", k, "records", idx[0], "..", idx[-1], len(idx), "first%64", idx[0] % 64, "last%64", idx[-1] % 64, "skipped in window", int(((soa.flag[idx] & 4) != 0).sum()),
          "span starts", idx[0] // 1024, idx[-1] // 1024)
# per-record GC
seq = soa.seq4.reshape(soa.n, 75)
hi, lo_n = seq >> 4, seq & 15
g = ((hi == 2) | (hi == 4)).sum(1) + ((lo_n == 2) | (lo_n == 4)).sum(1)
ok = (soa.tid >= 0) & ((soa.flag & 4) == 0)
print("gc of last 4 records", g[-4:], ok[-4:], int((g[-4:] * ok[-4:]).sum()), "last 8..4", int((g[-8:-4] * ok[-8:-4]).sum()), "records 503232..503295", int((g[503232:503296] * ok[503232:503296]).sum()))
import copy
for n in (soa.n - 4, soa.n - 3, soa.n - 68, soa.n - 1):
    class V: pass
    u = V(); u.refs = v.refs
    for k in ("tid", "pos", "flag", "l_qseq"):
        setattr(u, k, np.ascontiguousarray(getattr(v, k)[:n]))
    u.seq_off = np.ascontiguousarray(v.seq_off[:n + 1]); u.seq4 = np.ascontiguousarray(v.seq4[:int(v.seq_off[n])])
    u.cigar_off = v.cigar_off[:n + 1]; u.cigar = v.cigar
    rc, off2, wb, wg, wl, wt, wn = orc.window_counts(u, W)
    bins, gc, ln, touched, nc = ctx.window_counts(u, off2, W)
    bad = np.nonzero(gc != wg)[0]
    print("n =", n, "n%64 =", n % 64, "wrong windows", bad, (gc[bad].astype(np.int64) - wg[bad].astype(np.int64)))
