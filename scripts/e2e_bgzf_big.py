"""fastq_count / fastq_trim on a bgzip-compressed FASTQ of 12 GB of text (four launches of eight 88 MB chunks: ~3 GB of text per
inflate launch, beyond one framing call's 2 GiB): the report and the trimmed text against the plain file's.
   python scripts/e2e_bgzf_big.py > gpurun_out/e2e_bgzf_big.txt"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import highperformancengs_amd as hp  # noqa: E402
import bench_extra  # noqa: E402
import c4  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
td = tempfile.mkdtemp(prefix="bgzf_big_")
ctx = hp.Context(0)
with open(os.path.join(td, "big.fq"), "wb") as f:
    for k in range(3):
        f.write(bench_extra._fastq_text(ctx, 13_000_000, 150, 40 + k).data)
ctx.close()
exe = c4.build_synth(td)
subprocess.check_call([exe, "--bgzip", os.path.join(td, "big.fq"), os.path.join(td, "big.fq.gz"), "15"])
print(f"big.fq {os.path.getsize(os.path.join(td, 'big.fq')) / 1e9:.1f} GB, big.fq.gz {os.path.getsize(os.path.join(td, 'big.fq.gz')) / 1e9:.1f} GB")
outs = {}
for name in ("big.fq", "big.fq.gz"):
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(BIN, "fastq_count"), name], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1"})
    print(f"fastq_count {name}: {time.perf_counter() - t0:.3f} s rc {p.returncode}", [l for l in p.stderr.decode().splitlines() if l.startswith('[hpn]')][-2:])
    outs[name] = p.stdout.replace(name.encode(), b"X")
print("reports identical:", outs["big.fq"] == outs["big.fq.gz"], outs["big.fq.gz"].decode().strip())
sizes = {}
for name in ("big.fq", "big.fq.gz"):
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", name, "-s", "5", "-e", "140", "-o", "t_" + name.replace(".", "_")], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(f"fastq_trim {name}: {time.perf_counter() - t0:.3f} s rc {p.returncode}")
a, b = os.path.join(td, "t_big_fq.trim.fastq"), os.path.join(td, "t_big_fq_gz.trim.fastq")
print("trimmed texts identical:", subprocess.run(["cmp", "-s", a, b]).returncode == 0, os.path.getsize(a), os.path.getsize(b))
subprocess.run(["rm", "-rf", td])
