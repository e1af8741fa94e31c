"""The C4-shaped BAM (25 hg38 contigs, 7e7 reads, ~10 GB) through bam2depth / bam_sliding_count with HPN_TIMING=1: where the wall goes."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import c4
td = tempfile.mkdtemp(prefix="c4t_", dir="/tmp")
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
t0 = time.time()
bam, prefix = c4.synth(td, "hg38.bam", tg, 15, soa=False)
print(f"synth {time.time() - t0:.1f} s, {os.path.getsize(bam) / 1e9:.2f} GB")
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
for tool, env in (("bam2depth", {"HPN_NGPU": "1"}), ("bam2depth", {}), ("bam_sliding_count", {"HPN_NGPU": "1"}), ("bam_sliding_count", {})):
    for rep in range(2):
        t0 = time.time()
        p = subprocess.run([os.path.join(BIN, tool), "-w", "20000", "-o", "o", "hg38.bam"], cwd=td, env={**os.environ, "HPN_TIMING": "1", **env},
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print(f"== {tool} {env} rep {rep}: {time.time() - t0:.3f} s rc {p.returncode}")
        if rep:
            print("\n".join(l[:200] for l in p.stderr.decode().split("\n") if "amdgpu" not in l))
