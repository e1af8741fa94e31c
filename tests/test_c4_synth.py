"""CPU: scripts/bam_synth.cpp's --targets mode and tests/c4.py (the checker of the hg38-shaped file test) are held to the
compiled reference on a small instance: the BAM + .bai the generator writes are read by the reference's own bam2depth /
bam_sliding_count (oracle/_ref, built from /root/reference by oracle/Makefile), and the oracle's answers computed from the
generator's SoA sidecar must be those bytes.  Skipped where oracle/_ref is absent."""
import os
import shutil
import subprocess

import pytest

import c4

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.skipif(not os.access(os.path.join(REF, "bam2depth"), os.X_OK), reason="oracle/_ref not built")
def test_generator_sidecar_and_checker_against_the_reference(tmp_path):
    tg = [("chr1", 300000, 40000), ("chrTiny", 16569, 3300), ("chrEmpty", 5000, 0), ("chrX", 120000, 9000)]
    # (a target without reads: bam2depth still prints its windows, bam_sliding_count leaves its row out)
    bam, prefix = c4.synth(str(tmp_path), "t.bam", [t for t in tg], 3)
    soa = c4.Soa(prefix, len(tg))
    assert soa.n == sum(r for _, _, r in tg)
    W = 2000
    subprocess.run([os.path.join(REF, "bam2depth"), "-w", str(W), "-o", "d", "t.bam"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bed = open(tmp_path / "t.bam.1.bedGraph", "rb").read()
    dep = open(tmp_path / "d.1.depth", "rb").read()
    ob, od = b"", b""
    for t, (name, tlen, _) in enumerate(tg):
        runs, bins = c4.oracle_depth_target(soa, tg, t, W)
        b, d = c4.oracle_target_text(name, tlen, W, runs, bins)
        ob, od = ob + b, od + d
    assert ob == bed and od == dep
    subprocess.run([os.path.join(REF, "bam_sliding_count"), "-w", str(W), "-o", "s", "t.bam"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert c4.oracle_window_report(soa, tg, W) == open(tmp_path / "s.txt", "rb").read()
