"""BAM tools on the C4-shaped file over HPN_BAM_ROUNDS (88 MB chunks per inflate launch): one round of the chip's 4,608 decoder
waves per launch ends with its slowest block; several rounds per launch average that out."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import c4
os.makedirs("/tmp/c4p", exist_ok=True)
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
bam, _ = c4.synth("/tmp/c4p", "hg38.bam", tg, 15, soa=False)
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
for tool in ("bam2depth", "bam_sliding_count"):
    for ngpu in ("1", None):
        for rounds in ("1", None):
            env = {**os.environ, "HPN_TIMING": "1"}
            if ngpu: env["HPN_NGPU"] = ngpu
            if rounds: env["HPN_BAM_ROUNDS"] = rounds
            best = 9e9
            for rep in range(2):
                t0 = time.time()
                p = subprocess.run([os.path.join(BIN, tool), "-w", "20000", "-o", "o", "hg38.bam"], cwd="/tmp/c4p", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                best = min(best, time.time() - t0)
            ing = [l for l in p.stderr.decode().split("\n") if "ingest +" in l]
            print(f"{tool} workers {ngpu or 'default'} chunks per launch {rounds or 'default'}: {best:.3f} s rc {p.returncode} {ing[0][:90] if ing else ''}", flush=True)
