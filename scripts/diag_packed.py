"""Probe: a small two-contig BAM in samtools' layout and in packed blocks (records across block boundaries) through bam_sliding_count, the ingest's HPN_TIMING lines side by side."""
import os, subprocess, sys, tempfile
sys.path.insert(0, "tests")
import c4
td = tempfile.mkdtemp()
tg = [("chr1", 30_000_000, 4_000_000), ("chr9", 12_000_000, 1_000_000)]
for label, env in (("aligned", None), ("packed", {"BAM_SYNTH_PACKED": "1"})):
    d = os.path.join(td, label); os.makedirs(d)
    bam, _ = c4.synth(d, "s.bam", tg, 8, soa=False, env=env)
    p = subprocess.run([os.path.abspath("highperformancengs_amd/bin/bam_sliding_count"), "-w", "20000", "-o", "s", "s.bam"], cwd=d,
                       env={**os.environ, "HPN_TIMING": "2", "HPN_NGPU": "1"}, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(label, p.returncode)
    print("\n".join(l for l in p.stderr.decode().splitlines() if "record index" in l or "ingest" in l))
