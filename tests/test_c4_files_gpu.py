"""GPU: BASELINE configs[3] as FILES -- a coordinate-sorted BAM over the 25 hg38 primary contigs (true lengths) through
the built `bam2depth` / `bam_sliding_count` binaries: ingest -> per-target loop -> writers (bam2depth.c:325-339,
bam_sliding_count.c:389-416), on one context and on three workers (HPN_NGPU=3: targets largest first / record batches
in turn), against the oracle run on the generator's own records (tests/c4.py, pinned to the reference binaries by
tests/test_c4_synth.py).

Coverage of the file: 30x on chr21 and chrM, 3x on the other 23 contigs (7.0e7 reads of 150 bp, ~10 GB of BAM; the
full 30x file would be 6.2e8 reads / ~90 GB, beyond the suite's time).  Every output byte is compared: the bedGraph
(~3 GB, target by target as it is read back), the depth file, out.txt; per target also the run count and
sum(len x depth) = the M bases of the records bam2depth keeps."""
import os
import subprocess

import numpy as np
import pytest

import c4

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")


def _cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:  # noqa: BLE001
        pass
    return n


def _check_depth_outputs(d, bam_name, out_prefix, soa, tg, W, oracle):
    """oracle[t] = (n_runs, mass, bedGraph bytes digest pieces) computed once; compare the files in directory d."""
    with open(os.path.join(d, bam_name + ".1.bedGraph"), "rb") as fb, open(os.path.join(d, out_prefix + ".1.depth"), "rb") as fd:
        for t, (name, tlen, _) in enumerate(tg):
            bed, dep = oracle(t)
            got = fb.read(len(bed))
            assert got == bed, f"bedGraph differs in {name}"
            assert fd.read(len(dep)) == dep, f"depth rows differ in {name}"
        assert fb.read(1) == b"" and fd.read(1) == b""


def test_small_instance_every_route(tmp_path):
    """Four targets (one tiny, one without reads) through both tools on 1 / 2 / 3 workers: quick, and the empty target's
    rows are where a per-target loop goes wrong first."""
    tg = [("chr1", 3_000_000, 400_000), ("chrM", 16569, 3300), ("chrEmpty", 70_000, 0), ("chr9", 1_200_000, 100_000)]
    bam, prefix = c4.synth(str(tmp_path), "s.bam", tg, 4)
    soa = c4.Soa(prefix, len(tg))
    W = 20000
    cache = {}

    def oracle(t):
        if t not in cache:
            runs, bins = c4.oracle_depth_target(soa, tg, t, W)
            cache[t] = c4.oracle_target_text(tg[t][0], tg[t][1], W, runs, bins)
        return cache[t]
    want_txt = c4.oracle_window_report(soa, tg, W)
    # (HPN_BAM_CHUNK / HPN_BAM_ROUNDS: small chunks, one / three / seven of them under an inflate launch -- blocks carried from
    # chunk to chunk inside a launch and across launches)
    for env in ({}, {"HPN_NGPU": "2"}, {"HPN_NGPU": "3"}, {"HPN_BAM_GPU": "0"}, {"HPN_BAM_CHUNK": "200000", "HPN_BAM_ROUNDS": "1"},
                {"HPN_BAM_CHUNK": "150000", "HPN_BAM_ROUNDS": "3"}, {"HPN_BAM_CHUNK": "70000", "HPN_BAM_ROUNDS": "7", "HPN_NGPU": "1"}):
        d = tmp_path / ("run" + "".join(env.values()))
        d.mkdir()
        os.symlink(bam, d / "s.bam"), os.symlink(bam + ".bai", d / "s.bam.bai")
        e = {**os.environ, **env, "HPN_TIMING": "1"}
        p = subprocess.run([os.path.join(BIN, "bam2depth"), "-w", str(W), "-o", "d", "s.bam"], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        _check_depth_outputs(d, "s.bam", "d", soa, tg, W, oracle)
        p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", str(W), "-o", "s", "s.bam"], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert open(d / "s.txt", "rb").read() == want_txt, env


def test_small_instance_packed_across_blocks(tmp_path):
    """The same four targets written htsjdk's way (records packed across BGZF blocks, bamio.repack_bam; a .bai whose virtual
    offsets point into blocks): ~2,000 blocks, records run across most block ends and -- with small launches -- from one launch
    into the next; one context, per-target workers (HPN_NGPU), batches in turn (bam_sliding_count gives those back to one stream)."""
    from highperformancengs_amd import bamio
    tg = [("chr1", 3_000_000, 400_000), ("chrM", 16569, 3300), ("chrEmpty", 70_000, 0), ("chr9", 1_200_000, 100_000)]
    bam, prefix = c4.synth(str(tmp_path), "s.bam", tg, 4)
    soa = c4.Soa(prefix, len(tg))
    W = 20000
    cache = {}

    def oracle(t):
        if t not in cache:
            runs, bins = c4.oracle_depth_target(soa, tg, t, W)
            cache[t] = c4.oracle_target_text(tg[t][0], tg[t][1], W, runs, bins)
        return cache[t]
    want_txt = c4.oracle_window_report(soa, tg, W)
    packed = str(tmp_path / "packed.bam")
    assert bamio.repack_bam(bam, packed, 65280, level=1) == soa.n
    for env in ({"HPN_NGPU": "1"}, {"HPN_NGPU": "3"}, {"HPN_NGPU": "1", "HPN_BAM_CHUNK": "200000", "HPN_BAM_ROUNDS": "1"},
                {"HPN_NGPU": "2", "HPN_BAM_CHUNK": "150000", "HPN_BAM_ROUNDS": "3"}, {"HPN_NGPU": "1", "HPN_BAM_AHEAD": "0", "HPN_BAM_CHUNK": "300000"}):
        d = tmp_path / ("prun" + "".join(env.values()))
        d.mkdir()
        os.symlink(packed, d / "s.bam"), os.symlink(packed + ".bai", d / "s.bam.bai")
        e = {**os.environ, **env, "HPN_TIMING": "1"}
        p = subprocess.run([os.path.join(BIN, "bam2depth"), "-w", str(W), "-o", "d", "s.bam"], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert b"GPU ingest" in p.stderr and b"abandoned" not in p.stderr and b"host ingest" not in p.stderr, p.stderr.decode()
        _check_depth_outputs(d, "s.bam", "d", soa, tg, W, oracle)
        p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", str(W), "-o", "s", "s.bam"], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert b"GPU ingest" in p.stderr and b"host ingest" not in p.stderr, p.stderr.decode()
        assert open(d / "s.txt", "rb").read() == want_txt, env


def test_hg38_shaped_bam_through_the_tools(tmp_path):
    tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
    n_reads = sum(r for _, _, r in tg)
    assert 6.9e7 < n_reads < 7.2e7
    bam, prefix = c4.synth(str(tmp_path), "hg38.bam", tg, max(2, _cpus() - 1))
    soa = c4.Soa(prefix, len(tg))
    assert soa.n == n_reads
    for ext in (".tid", ".pos", ".flag", ".kind", ".seq4"):      # the arrays are in memory now
        os.unlink(prefix + ext)
    W = 20000
    m_per = np.array([150, 148, 150, 140], np.int64)[soa.kind]   # M bases per record of the four CIGAR kinds
    kept = (soa.flag & 0x704) == 0
    runs_total = 0
    summary = {}

    # ---- bam2depth, one context: every byte, and per target the run count and the coverage mass --------------------
    d1 = tmp_path / "one"
    d1.mkdir()
    os.symlink(bam, d1 / "hg38.bam"), os.symlink(bam + ".bai", d1 / "hg38.bam.bai")
    # (HPN_NGPU=1: one context, which is also what bam2depth takes by itself on one device)
    p = subprocess.run([os.path.join(BIN, "bam2depth"), "-w", str(W), "-o", "d", "hg38.bam"], cwd=d1, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_NGPU": "1"})
    assert p.returncode == 0, p.stderr.decode()
    with open(d1 / "hg38.bam.1.bedGraph", "rb") as fb, open(d1 / "d.1.depth", "rb") as fd:
        for t, (name, tlen, _) in enumerate(tg):
            runs, bins = c4.oracle_depth_target(soa, tg, t, W)
            lo, hi = int(soa.lo[t]), int(soa.lo[t + 1])
            mass = int(m_per[lo:hi][kept[lo:hi]].sum())
            assert int(((runs[:, 1] - runs[:, 0]).astype(np.int64) * runs[:, 2]).sum()) == mass == int(round(bins.sum()))
            bed, dep = c4.oracle_target_text(name, tlen, W, runs, bins)
            assert bed.count(b"\n") == len(runs)
            got = fb.read(len(bed))
            assert got == bed, f"bedGraph differs in {name}"
            assert fd.read(len(dep)) == dep, f"depth rows differ in {name}"
            summary[name] = (len(runs), mass)
            runs_total += len(runs)
            del runs, bins, bed, dep, got
        assert fb.read(1) == b"" and fd.read(1) == b""
    assert runs_total > 3.0e7                                     # (evenly spaced starts: a read start often meets a read end; random starts would give ~1.2e8)
    # ---- bam2depth over three workers: the same files ------------------------------------------------------------------
    d3 = tmp_path / "three"
    d3.mkdir()
    os.symlink(bam, d3 / "hg38.bam"), os.symlink(bam + ".bai", d3 / "hg38.bam.bai")
    p = subprocess.run([os.path.join(BIN, "bam2depth"), "-w", str(W), "-o", "d", "hg38.bam"], cwd=d3, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_NGPU": "3", "HPN_TIMING": "1"})
    assert p.returncode == 0, p.stderr.decode()
    assert b"GPU ingest on 3 workers" in p.stderr and b"abandoned" not in p.stderr, p.stderr.decode()
    for f in ("hg38.bam.1.bedGraph", "d.1.depth"):
        assert subprocess.run(["cmp", "-s", str(d1 / f), str(d3 / f)]).returncode == 0, f
        os.unlink(d3 / f)
    os.unlink(d1 / "hg38.bam.1.bedGraph")
    # ---- bam_sliding_count: out.txt on one context and on three workers --------------------------------------------------
    want = c4.oracle_window_report(soa, tg, W)
    for d, env in ((d1, {}), (d3, {"HPN_TIMING": "1", "HPN_NGPU": "3"})):        # the default on one device (one worker); batches to three workers in turn
        p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", str(W), "-o", "s", "hg38.bam"], cwd=d, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env={**os.environ, **env})
        assert p.returncode == 0, p.stderr.decode()
        assert open(d / "s.txt", "rb").read() == want, env
        if "HPN_TIMING" in env:
            assert b"GPU ingest on 3 workers" in p.stderr and b"abandoned" not in p.stderr, p.stderr.decode()
    rows = want.split(b"\n")
    assert len(rows) == 1 + 25 + 1 and rows[1].startswith(b"chr1\t248956422\t")


def test_gc_window_beyond_the_float32_domain_is_refused(tmp_path):
    """A window whose G/C sum reaches 2^24: the reference's float32 accumulation (bam_sliding_count.c:119-121) is order-
    dependent there (pinned against the reference in tests/test_c4_synth.py); the tool stops with the domain error's exit
    code instead of printing other digits, on every route; a window size that keeps the sums exact gives the oracle's bytes."""
    tg = [("chrBig", 2_000_000, 260_000), ("chrS", 50_000, 1000)]
    bam, prefix = c4.synth(str(tmp_path), "g.bam", tg, 4)
    soa = c4.Soa(prefix, len(tg))
    for env in ({}, {"HPN_NGPU": "2"}, {"HPN_BAM_GPU": "0"}):
        d = tmp_path / ("r" + "".join(env.values()))
        d.mkdir()
        os.symlink(bam, d / "g.bam"), os.symlink(bam + ".bai", d / "g.bam.bai")
        e = {**os.environ, **env}
        p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", "3000000", "-o", "s", "g.bam"], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 2 and b"order-dependent" in p.stderr and not os.path.exists(d / "s.txt"), p.stderr.decode()
        p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", "100000", "-o", "s", "g.bam"], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert open(d / "s.txt", "rb").read() == c4.oracle_window_report(soa, tg, 100000)
