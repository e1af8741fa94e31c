"""BASELINE configs[3] as FILES: a coordinate-sorted BAM over the 25 hg38 primary contigs (true lengths) made by
scripts/bam_synth.cpp (SURVEY.md 8d's read / CIGAR / flag mix), its records kept beside it as raw SoA files, and the
oracle's answers for bam2depth (bam2depth.c:325-339: per-target loop -> bedGraph + depth) and bam_sliding_count
(bam_sliding_count.c:389-416 -> out.txt) computed from those records.  Shared by tests/test_c4_files_gpu.py and
bench_extra.py's C4 leg; test infrastructure (uses oracle/)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HG38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
        ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
        ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189),
        ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
        ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415), ("chrM", 16569)]

KIND_CIGAR = [[150 << 4], [40 << 4, (2 << 4) | 1, 108 << 4], [60 << 4, (5 << 4) | 2, 90 << 4], [(10 << 4) | 4, 140 << 4]]


def targets(depth_of):
    """[(name, len, reads)] with reads = depth * len / 150; depth_of(name, len) -> coverage."""
    return [(n, l, max(1, int(depth_of(n, l) * l / 150))) for n, l in HG38]


def build_synth(workdir):
    # (the generator is BUILT where programs may run: a work directory under /dev/shm is usually mounted noexec)
    import tempfile
    if os.path.realpath(workdir).startswith("/dev/shm"):
        workdir = os.path.join(tempfile.gettempdir(), "hpn_c4_exe")
        os.makedirs(workdir, exist_ok=True)
    exe = os.path.join(workdir, "bam_synth")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "scripts", "bam_synth.cpp"), "-o", exe, "-lz", "-lpthread"])
    return exe


def synth(workdir, name, tg, threads, soa=True, env=None):
    """-> path of the BAM (+ .bai) and the SoA prefix."""
    exe = build_synth(workdir)
    bam = os.path.join(workdir, name)
    spec = ",".join(f"{n}:{l}:{r}" for n, l, r in tg)
    cmd = [exe, bam, "--targets", spec, str(threads)] + ([bam + ".soa"] if soa else [])
    subprocess.check_call(cmd, env={**os.environ, **(env or {})})
    return bam, bam + ".soa"


class Soa:
    def __init__(self, prefix, n_targets):
        self.tid = np.fromfile(prefix + ".tid", np.int32)
        self.pos = np.fromfile(prefix + ".pos", np.int32)
        self.flag = np.fromfile(prefix + ".flag", np.uint32)
        self.kind = np.fromfile(prefix + ".kind", np.uint8)
        self.seq4 = np.fromfile(prefix + ".seq4", np.uint8)
        self.n = len(self.tid)
        assert len(self.seq4) == 75 * self.n
        ncig = np.array([len(k) for k in KIND_CIGAR], np.uint32)[self.kind]
        self.cigar_off = np.zeros(self.n + 1, np.uint32)
        np.cumsum(ncig, out=self.cigar_off[1:])
        table = np.zeros((4, 3), np.uint32)
        for k, ops in enumerate(KIND_CIGAR):
            table[k, :len(ops)] = ops
        words = table[self.kind]
        self.cigar = words[np.arange(3)[None, :] < ncig[:, None]].astype(np.uint32)
        self.l_qseq = np.full(self.n, 150, np.int32)
        self.seq_off = (np.arange(self.n + 1, dtype=np.uint64) * 75)
        # records are sorted by target: slice [lo[t], lo[t+1])
        self.lo = np.searchsorted(self.tid, np.arange(n_targets + 1))


class _View:
    """What tests/orc.py's wrappers take: refs + SoA arrays (here: one target's slice, or everything)."""


def target_view(soa, tg, t):
    lo, hi = int(soa.lo[t]), int(soa.lo[t + 1])
    v = _View()
    v.refs = [(n, l) for n, l, _ in tg]
    v.tid = np.ascontiguousarray(soa.tid[lo:hi])
    v.pos = np.ascontiguousarray(soa.pos[lo:hi])
    v.flag = np.ascontiguousarray(soa.flag[lo:hi])
    v.cigar_off = (soa.cigar_off[lo:hi + 1] - soa.cigar_off[lo]).astype(np.uint32)
    v.cigar = np.ascontiguousarray(soa.cigar[int(soa.cigar_off[lo]):int(soa.cigar_off[hi])])
    if len(v.cigar) == 0:
        v.cigar = np.zeros(1, np.uint32)
    return v


def whole_view(soa, tg):
    v = _View()
    v.refs = [(n, l) for n, l, _ in tg]
    for k in ("tid", "pos", "flag", "l_qseq", "seq_off", "seq4", "cigar_off", "cigar"):
        setattr(v, k, getattr(soa, k))
    return v


def oracle_depth_target(soa, tg, t, W, mask=0x704):
    """orc_depth_target (bam2depth.c:86-110,203-236,132-176) over target t's records -> (runs (n,3) int32, bins f64)."""
    import orc
    rc, runs, bins = orc.depth_target(target_view(soa, tg, t), t, W, mask)
    assert rc == 0
    return runs, bins


def oracle_target_text(name, tlen, W, runs, bins):
    """-> (bedGraph lines, depth rows) of one target, as hash2BedGraph's fprintf (:217) and output_bins (:238-246) write them."""
    import orc
    L = orc.lib()
    fb, fd = orc._CFile(), orc._CFile()
    arr = np.ascontiguousarray(runs, np.int32)
    L.orc_fmt_bedgraph(fb.fp, name.encode(), C.cast(arr.ctypes.data, C.POINTER(orc.Run)), len(arr))
    L.orc_fmt_depth_bins(fd.fp, name.encode(), tlen, W, bins)
    return fb.read(), fd.read()


def oracle_window_report(soa, tg, W):
    """fetch_func + cal_GC + calc_winGC + output_count_GC (bam_sliding_count.c:84-164) -> the bytes of out.txt."""
    import orc
    return orc.window_report(whole_view(soa, tg), W)
