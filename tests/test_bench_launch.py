"""CPU: `bench.py --gpus N` makes N ranks or fails loudly -- it never prints an n_gpus it did not run.
The driver's command is `python3 bench.py --gpus N ...` with no launcher around it; round 3's bench.py parsed --gpus and never read
it, so N = 8 would have produced ONE rank and `n_gpus: 1`.  Here the launcher and the rank arithmetic run with the test-only CPU
backend (tests/stub/bench_backend.py: gloo, tallies from the oracle).  Reference seam: the workers reduceStats sums
(fastq_count_kthread.c:180-210, :270)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, "tests")


def _bench(args, env=None, launcher_env=True):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["PYTHONPATH"] = TESTS + os.pathsep + e.get("PYTHONPATH", "")
    e["HPN_BENCH_BACKEND"] = "stub.bench_backend:Backend"
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


COMMON = ["--steps", "2", "--warmup", "1", "--reads", "3000", "--read-len", "60", "--no-extra", "--no-cpu-baseline"]


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks_and_says_so(n):
    p = _bench(["--gpus", str(n)] + COMMON)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["config"]["ranks"] == n and j["config"]["rccl_ranks"] == n
    assert j["config"]["launcher"] == "bench.py" and j["config"]["allreduce"] == "rccl-native"
    assert len(j["roofline"]["kernel_ms_per_rank"]) == n     # every rank reported its own kernel time
    assert j["scaling"] == "weak" and j["config"]["reads_per_gpu"] == 3000
    assert j["config"]["backend"].startswith("stub")          # and nobody can mistake this line for a measurement
    # whole-job value: N ranks x reads x length x steps over the slowest rank's time
    # (value is printed with three decimals: on a loaded host it is small enough for that rounding to matter)
    assert abs(j["value"] - n * 3000 * 60 * 2 / (j["ms_per_step"] * 2e-3) / 1e9) < 1e-2 * j["value"] + 6e-4


def test_more_ranks_than_devices_fails_loudly():
    p = _bench(["--gpus", "2"] + COMMON, env={"HPN_STUB_DEVICES": "1"})
    assert p.returncode != 0
    assert b"2 ranks need 2 devices, this node has 1" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_world_size_that_disagrees_with_gpus_is_refused():
    p = _bench(["--gpus", "4"] + COMMON, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and b"--gpus 4 but the launcher started WORLD_SIZE=1" in p.stderr
    assert p.stdout.strip() == b""


def test_a_failing_rank_takes_the_job_down_without_a_line():
    # rank 1 dies at start-up (its backend cannot be imported): the launcher stops the others and exits non-zero
    code = "import os, sys\nif os.environ.get('RANK') == '1': raise SystemExit(7)\nfrom stub.bench_backend import *\n"
    d = os.path.join(TESTS, "stub")
    path = os.path.join(d, "_dying_backend.py")
    open(path, "w").write(code)
    try:
        p = _bench(["--gpus", "2", "--rank-timeout", "120"] + COMMON, env={"HPN_BENCH_BACKEND": "stub._dying_backend:Backend"})
    finally:
        os.unlink(path)
    assert p.returncode == 1 and b"rank 1 of 2 failed" in p.stderr
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_torchrun_launch_still_works():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e["PYTHONPATH"] = TESTS + os.pathsep + e.get("PYTHONPATH", "")
    e["HPN_BENCH_BACKEND"] = "stub.bench_backend:Backend"
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    j = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["launcher"] == "torchrun"
