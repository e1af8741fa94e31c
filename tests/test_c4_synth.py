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


@pytest.mark.skipif(not os.access(os.path.join(REF, "bam_sliding_count"), os.X_OK), reason="oracle/_ref not built")
def test_float32_gc_sum_beyond_2_pow_24_is_order_dependent_and_the_oracle_replays_it(tmp_path):
    """bam_sliding_count.c:121 adds an unsigned short to a float per record: past 2^24 every += rounds, so the window's value
    depends on the order of the records.  The oracle replays the additions in record order (orc_window_gc_f32) and must print the
    reference's digits; a sum taken as an integer and converted prints others -- which is why the tool refuses such a window
    (tests/test_c4_files_gpu.py::test_gc_window_beyond_the_float32_domain_is_refused)."""
    import numpy as np
    import orc
    tg = [("chrBig", 2_000_000, 260_000), ("chrS", 50_000, 1000)]
    bam, prefix = c4.synth(str(tmp_path), "g.bam", tg, 3)
    soa = c4.Soa(prefix, len(tg))
    W = 3_000_000
    subprocess.run([os.path.join(REF, "bam_sliding_count"), "-w", str(W), "-o", "s", "g.bam"], cwd=tmp_path, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = open(tmp_path / "s.txt", "rb").read()
    assert c4.oracle_window_report(soa, tg, W) == ref
    rc, off, bins, gc, ln, touched, _ = orc.window_counts(c4.whole_view(soa, tg), W)
    assert rc == 0 and int(gc[0]) >= 1 << 24                       # the window is past the exact range ...
    naive = np.float32(int(gc[0])) / np.float32(ln[0]) * np.float32(100)
    assert ("%f" % naive).encode() not in ref.split(b"\n")[1]      # ... and the converted integer sum would print other digits
