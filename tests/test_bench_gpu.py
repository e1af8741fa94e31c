"""GPU: what bench.py does on the box it is given -- before the driver's first real 8-GPU run.
  * `--gpus 2` on a ONE-GPU box: every rank stops with "2 ranks need 2 devices", the launcher passes the refusal on as exit code 2,
    and no JSON line appears (a line with n_gpus = 2 from one device would be a fabricated scaling point);
  * `--gpus 1`, a small resident batch: ONE line, n_gpus 1, the collective fields empty, the roofline object complete.
The N-rank arithmetic itself runs on CPU over gloo (tests/test_bench_launch.py, tests/test_shard_gloo.py).  Reference seam of the
sum: reduceStats (fastq_count_kthread.c:180-210)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--reads", "2e5", "--no-extra", "--no-cpu-baseline"]


def _bench(args):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HPN_BENCH_BACKEND")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)


def test_two_ranks_on_a_one_gpu_box_are_refused_without_a_line():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("a one-GPU box is what this is about")
    p = _bench(["--gpus", "2"] + SMALL)
    assert p.returncode == 2, (p.returncode, p.stderr.decode()[-2000:])
    assert b"2 ranks need 2 devices, this node has 1" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_one_rank_prints_one_complete_line():
    p = _bench(["--gpus", "1"] + SMALL)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["ranks"] == 1 and j["config"]["rccl_ranks"] is None and j["config"]["launcher"] == "none"
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert len(r["kernel_ms_per_rank"]) == 1 and j["unit"] == "Gbases/s" and j["dtype"] == "u8"
