"""fastq_trim on ONE plain 16.3 GB FASTQ -> t.trim.fastq (the bench's steady-state trim leg), with the tool's own timing lines and
the knobs of its host side: lanes (HPN_NGPU).  (Round 4 also swept writer threads -- HPN_WRITE_THREADS -- and the piece size;
the writer is one thread since round 5: host/text_stream.hpp write_slab.)
   python scripts/e2e_trim.py > gpurun_out/e2e_trim_r04.txt"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import highperformancengs_amd as hp  # noqa: E402
import bench_extra  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
HOOKS = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")      # (HPN_TRIM_NOWRITE is a test-hooks switch: host/knobs.hpp)
td = tempfile.mkdtemp(prefix="e2e_trim_")
ctx = hp.Context(0)
with open(os.path.join(td, "big.fq"), "wb") as fb:
    for k in range(4):
        blk = bench_extra._fastq_text(ctx, 13_000_000, 150, 40 + k)
        fb.write(blk.data)
        del blk
ctx.close()
torch.cuda.empty_cache()
print(f"big.fq: {os.path.getsize(os.path.join(td, 'big.fq')) / 1e9:.1f} GB")
sizes = set()
for env in ({}, {"HPN_NGPU": "1"}, {"HPN_NGPU": "3"},
            {"HPN_TRIM_NOWRITE": "1"}):
    best, err = 1e9, ""
    for _ in range(2):
        out = os.path.join(td, "t.trim.fastq")
        if os.path.exists(out):
            os.unlink(out)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(HOOKS if "HPN_TRIM_NOWRITE" in env else BIN, "fastq_trim"), "-i", "big.fq", "-s", "5", "-e", "140", "-o", "t"], cwd=td,
                           env={**os.environ, "HPN_TIMING": "1", **env}, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        if dt < best:
            best, err = dt, p.stderr.decode()
    sizes.add(os.path.getsize(os.path.join(td, "t.trim.fastq")))
    print(f"--- fastq_trim {env}: {best:.3f} s")
    for l in err.splitlines():
        if l.startswith("[hpn") or l.startswith("Finished"):
            print("    " + l[:240])
    sys.stdout.flush()
print("output sizes:", sizes)
subprocess.run(["rm", "-rf", td])
