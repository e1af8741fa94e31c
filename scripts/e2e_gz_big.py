"""A large single-member .fastq.gz (pigz-style: raw deflate pieces, one trailer) through fastq_count with HPN_TIMING: where the
gzip route's wall goes (block-start search, upload, device inflate, framing + tally, CRC-32)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import highperformancengs_amd as hp
import bench_extra

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 13_000_000
ctx = hp.Context(0)
raw = bench_extra._fastq_text(ctx, n, 150, 40).tobytes()
ctx.close()
t0 = time.time()
blob = bench_extra._gz_single_member(raw, 256, 16)
print(f"text {len(raw) / 1e9:.2f} GB -> gzip {len(blob) / 1e9:.2f} GB in {time.time() - t0:.1f} s")
with open("/tmp/big.fq.gz", "wb") as f:
    f.write(blob)
del raw, blob
exe = os.path.join(ROOT, "highperformancengs_amd", "bin", "fastq_count")
for env in ({}, {"HPN_GZ_DEBUG": "1"}, {"HPN_GZ_DEBUG": "1", "HPN_GZ_CRC": "0"}):
    t0 = time.time()
    p = subprocess.run([exe, "/tmp/big.fq.gz"], env={**os.environ, "HPN_TIMING": "1", **env}, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(env, f"{time.time() - t0:.3f} s")
    print(p.stderr.decode().strip())
    print(p.stdout.decode().strip())
