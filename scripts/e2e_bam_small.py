"""The 605 MB BAM of the bench's e2e legs (4e6 reads over 2 x 10 Mb) through the BAM tools over HPN_BAM_ROUNDS."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import c4
os.makedirs("/tmp/c4s", exist_ok=True)
tg = [("chrA", 10_000_000, 2_000_000), ("chrB", 10_000_000, 2_000_000)]
bam, _ = c4.synth("/tmp/c4s", "s.bam", tg, 15, soa=False)
print(os.path.getsize(bam) / 1e6, "MB")
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
for tool in ("bam2depth", "bam_sliding_count"):
    for rounds in ("1", "2", "4"):
        ts = []
        for rep in range(4):
            t0 = time.time()
            p = subprocess.run([os.path.join(BIN, tool), "-o", "o", "s.bam"], cwd="/tmp/c4s", env={**os.environ, "HPN_TIMING": "1", "HPN_BAM_ROUNDS": rounds}, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            ts.append(time.time() - t0)
        print(tool, "rounds", rounds, " ".join(f"{t:.3f}" for t in ts), [l[:110] for l in p.stderr.decode().split("\n") if "hpn]" in l][-2:], flush=True)
