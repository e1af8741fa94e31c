"""The C4-shaped BAM (25 hg38 contigs, 7e7 reads, ~10 GB) through bam2depth / bam_sliding_count with HPN_TIMING=1: where the wall goes."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import c4
os.makedirs("/tmp/c4p", exist_ok=True)
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
bam, prefix = c4.synth("/tmp/c4p", "hg38.bam", tg, 15, soa=False)
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
for tool, env in (("bam2depth", {}), ("bam_sliding_count", {})):
    for rep in range(2):
        t0 = time.time()
        p = subprocess.run([os.path.join(BIN, tool), "-w", "20000", "-o", "o", "hg38.bam"], cwd="/tmp/c4p", env={**os.environ, "HPN_TIMING": "1", **env},
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print(f"== {tool} {env} rep {rep}: {time.time() - t0:.3f} s rc {p.returncode}")
        if rep:
            print("\n".join(l[:200] for l in p.stderr.decode().split("\n") if "amdgpu" not in l and not (l.startswith("chr") and l[3:5] not in ("1 ", "M ", "21"))))
