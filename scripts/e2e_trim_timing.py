"""fastq_trim -s 5 -e 140 on a plain 2.5 GB / 8e6-read file with HPN_TIMING: reader / writer waits against device time."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import highperformancengs_amd as hp
import bench_extra
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8_000_000
ctx = hp.Context(0)
raw = bench_extra._fastq_text(ctx, n, 150, 40).tobytes()
ctx.close()
open("/tmp/t.fq", "wb").write(raw)
exe = os.path.join(ROOT, "highperformancengs_amd", "bin", "fastq_trim")
for rep in range(3):
    t0 = time.time()
    p = subprocess.run([exe, "-i", "/tmp/t.fq", "-o", "/tmp/t_out", "-s", "5", "-e", "140"], env={**os.environ, "HPN_TIMING": "1"}, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(f"rep {rep}: {time.time() - t0:.3f} s")
    print("\n".join(l for l in p.stderr.decode().split("\n") if "amdgpu" not in l)[:600])
