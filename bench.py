#!/usr/bin/env python3
"""bench.py -- fastq_count hot path on MI355X: Gbases/s and % of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` IS the number of ranks.  Under a launcher (WORLD_SIZE in the environment) the rank checks WORLD_SIZE == N and
stops otherwise; without one and N > 1 this process becomes the launcher: it starts N fresh rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), never touches a GPU itself, relays rank 0's JSON line and exits non-zero when any
rank fails or the job times out.  N ranks need N devices: on a smaller node every rank stops with that message.

The headline: a step = one pass of hpn_fastq_tally over the rank's resident batch of synthetic
reads (BASELINE.json configs[1]: 1e9 x 150 bp per GPU, generated in HBM by the
counter-based generator), plus -- for N > 1 -- the one sum all-reduce of the
count vector, plus the fetch of the counts to the host.  Inputs are resident in
HBM when the timed region starts (inflate / PCIe are host work).
Weak scaling: every rank holds its own 1e9-read shard (configs[4]: 8e9 reads on 8 GPUs).
Rank 0 prints ONE JSON line.  Its `extra` object (bench_extra.py) holds what the headline does not: the exact check of
the headline launch, the kernels of the other named configurations (K1L, K2, K3 + K4, K5) with their own
roofline fractions, and end-to-end legs through the CLI binaries with the reference binary timed beside each.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s measured achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=float, default=1e9, help="reads per GPU (BASELINE configs[1])")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--full-matrix", action="store_true", help="also build Quality[128][512] (kthread -L path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline only: skip the exact check, the other kernels and the end-to-end legs")
    ap.add_argument("--no-ragged", action="store_true", help="skip the mixed-length K1 run behind the timed region (PMC passes: every k_tally_scan launch is then the headline's)")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--rank-timeout", type=float, default=3000.0, help="launcher: seconds the N ranks may take together")
    return ap.parse_args()


# --------------------------------------------------------------------------------------
# --gpus N without a launcher: this process starts the N ranks (and does nothing else)
# --------------------------------------------------------------------------------------
def launch_ranks(a):
    """N fresh processes, one per GPU, each running this file as a rank; stands where reduceStats' caller starts its
    workers (fastq_count_kthread.c:270, klib/kthread.c:48).  No exec, no GPU call in this process."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HPN_BENCH_LAUNCHER="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # rank 0's pipe never fills
    reader.start()
    deadline = time.time() + a.rank_timeout
    failed = None
    live = set(range(a.gpus))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, f"exit code {rc}")
                break
        if live and failed is None:
            if time.time() > deadline:
                failed = (min(live), f"still running after {a.rank_timeout:.0f} s")
            else:
                time.sleep(0.2)
    if failed is not None:
        refused = procs[failed[0]].returncode == 2     # (a rank that REFUSED to run -- too few devices, a launcher that disagrees -- : its code is the job's)
        for r in live:   # exactly the processes started above, by pid
            procs[r].kill()
        for pr in procs:
            try:
                pr.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
        print(f"[bench] rank {failed[0]} of {a.gpus} failed ({failed[1]}): no result", file=sys.stderr)
        return 2 if refused else 1
    reader.join(timeout=30)
    lines = [l for l in (out0[0].decode() if out0 else "").splitlines() if l.startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


class HipBackend:
    """Where the ranks' buffers live and what makes their contexts: torch for device memory and torch.distributed, the
    library's hpn_ctx for everything else.  (tests/stub/bench_backend.py is the CPU counterpart the launcher test injects
    with HPN_BENCH_BACKEND; it is never used otherwise.)"""
    name, device, dist_backend = "hip", "cuda", "nccl"

    def __init__(self):
        import torch
        self.torch = torch

    def n_devices(self):
        return self.torch.cuda.device_count()   # (counting devices does not initialise the GPU)

    def open(self, local):
        import highperformancengs_amd as hp
        if not self.torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        self.torch.cuda.set_device(local)
        self.local = local
        self.ctx = hp.Context(local)
        return self.ctx

    def dist_kwargs(self):
        return {"device_id": self.torch.device("cuda", self.local)}

    def unique_id(self):
        import highperformancengs_amd as hp
        return hp.comm_unique_id()

    def resident_batch(self, n, L, rank):
        from highperformancengs_amd import shard
        torch = self.torch
        free, _total = torch.cuda.mem_get_info()
        need = n * (L + 8) + (1 << 30)
        if need > free * 0.92:  # smaller HBM than expected: shrink the resident batch, say so
            n = int(free * 0.92 - (1 << 30)) // (L + 8)
        first = shard.weak_shard_first(rank, n)  # global record index of this shard (counter-based generator)
        d_qual = torch.empty(n * L, dtype=torch.uint8, device="cuda")
        d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        self.ctx.synth_fastq_dev(12345, first, n, L, d_qual, None, d_off)
        self.ctx.sync()
        return d_qual, d_off, n, first

    def sync(self):
        self.ctx.sync()
        self.torch.cuda.synchronize()

    def rccl_ranks(self):
        try:
            return self.ctx.comm_count()
        except Exception:  # noqa: BLE001
            return None


def make_backend():
    spec = os.environ.get("HPN_BENCH_BACKEND")
    if not spec:
        return HipBackend()
    import importlib
    mod, _, attr = spec.partition(":")
    return getattr(importlib.import_module(mod), attr or "Backend")()


# --------------------------------------------------------------------------------------
# CPU baseline leg: the only place bench.py touches oracle/ (reported, never the product)
# --------------------------------------------------------------------------------------
def usable_cpus():
    """CPUs this process can keep busy: affinity mask cut down to the cgroup CPU quota (the GPU
    boxes show 256 online CPUs under a 16-CPU quota; 256 threads there are 16 cores' worth)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:  # noqa: BLE001
            pass
    return n


def cpu_baseline(read_len, target_s):
    orc_so = os.path.join(ROOT, "oracle", "liborc.so")
    if not os.path.exists(orc_so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liborc.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(orc_so)
    L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_counts_new.restype = C.c_void_p
    L.orc_counts_free.argtypes = [C.c_void_p]
    L.orc_count_files_threaded.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
    cores = usable_cpus()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "fastq_count_kthread")
    kind = "reference" if os.access(ref_bin, os.X_OK) else "port"
    td = tempfile.mkdtemp(prefix="hpn_cpu_")
    # one plain-text shard per core (the reference parallelises per file), bounded to 8 GB
    # and to a quarter of the free space of the temp directory
    budget = min(8 << 30, shutil.disk_usage(td).free // 4)
    per_shard = int(max(10_000, min(2_000_000, budget // (cores * (2 * read_len + 16)))))
    try:
        paths = [os.path.join(td, f"shard{i}.fq") for i in range(cores)]
        with ThreadPoolExecutor(cores) as ex:  # the C writer releases the GIL
            list(ex.map(lambda i: L.orc_synth_write_fastq(paths[i].encode(), 12345, i * per_shard, per_shard,
                                                          read_len, read_len, 0), range(cores)))

        def run(reps):
            files = [p for p in paths for _ in range(reps)]
            if kind == "reference":
                t0 = time.perf_counter()
                subprocess.run([ref_bin, "-t", str(cores), "-o", os.path.join(td, "merged.tsv")] + files,
                               cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                dt = time.perf_counter() - t0
                for f in os.listdir(td):
                    if f.endswith(".tsv"):
                        os.unlink(os.path.join(td, f))
                return dt
            arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
            merged = L.orc_counts_new()
            sec = C.c_double()
            rc = L.orc_count_files_threaded(arr, len(files), cores, merged, C.byref(sec))
            L.orc_counts_free(merged)
            assert rc == 0
            return sec.value

        t1 = run(1)  # also warms the page cache
        reps = max(1, min(64, int(target_s / max(t1, 1e-3))))
        dt = run(reps)
        bases = cores * reps * per_shard * read_len
        return {"value": round(bases / dt / 1e9, 4), "unit": "Gbases/s", "cores": cores, "kind": kind,
                "sample": f"{cores} plain FASTQ shards x {per_shard} reads x {read_len} bp, each listed {reps}x "
                          f"({bases / 1e9:.2f} Gbases, {dt:.2f} s wall), fastq_count_kthread -t {cores} one thread per file"}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a {a.gpus}-GPU number "
              f"from {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    import torch  # noqa: F401
    import torch.distributed as dist
    import highperformancengs_amd as hp  # noqa: F401
    from highperformancengs_amd import _lib, shard  # noqa: F401

    be = make_backend()
    have = be.n_devices()
    if have < world:   # every rank sees the same count and stops here, before any rendezvous
        print(f"bench.py: {world} ranks need {world} devices, this node has {have} (one process per GPU; "
              f"ranks never share a device)", file=sys.stderr)
        sys.exit(2)
    ctx = be.open(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(be.dist_backend, rank=rank, world_size=world, **be.dist_kwargs())

    # ---- resident batch: this rank's shard of the N x 1e9-read job -------------------
    L = a.read_len
    d_qual, d_off, n, first = be.resident_batch(int(a.reads), L, rank)

    # ---- the one collective: native RCCL on the context's stream, else torch.distributed (shard.ShardedTally) --
    job = shard.ShardedTally(ctx, rank, world, device=be.device, full_matrix=a.full_matrix)
    allreduce = job.setup(be.unique_id)
    if allreduce == "torch.distributed" and rank == 0:
        print(f"[bench] native RCCL init failed on some rank ({getattr(job, 'why', 'another rank')}); using torch.distributed",
              file=sys.stderr)
    from highperformancengs_amd import api as _api
    rccl_lib = _api.comm_library() if be.name == "hip" and (allreduce.startswith("rccl") or world == 1) else None   # which librccl the native binding resolved to
    # ranks in the communicator the sum goes through: ncclCommCount of the native one, else torch.distributed's group (backend
    # "nccl" IS RCCL).  EVERY rank's figure must be `world`: a number from fewer ranks than --gpus is never printed.
    rccl_ranks = be.rccl_ranks() if allreduce.startswith("rccl") else (dist.get_world_size() if world > 1 else None)
    if world > 1:
        seen = [int(x) for x in shard.gather_floats(float(rccl_ranks or 0), be.device)]
        if any(x != world for x in seen):
            if rank == 0:
                print(f"bench.py: the collective spans {seen} ranks (per rank) where --gpus {world} asked for {world}: no result", file=sys.stderr)
            sys.exit(3)
    kernel_ms = []

    def step():
        out = job.step(d_qual, d_off, n)
        kernel_ms.append(ctx.last_kernel_ms(0))
        return out

    def fence():
        be.sync()
        if world > 1:
            dist.barrier()
        be.sync()

    for _ in range(a.warmup):
        out = step()
    kernel_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = shard.max_over_ranks(time.perf_counter() - t0, be.device)
    k_mine = sum(kernel_ms) / max(1, len(kernel_ms))
    k_ranks = shard.gather_floats(k_mine, be.device)   # every rank's mean kernel time, in rank order

    # ---- the same kernel on reads of MIXED lengths (after the timed region; rank 0's figure is reported) --------------------
    # The headline's reads are all 150 bp, which lets K1's offset pass add 1,024 equal lengths at once; trimmed data does not.
    # Same resident quality bytes, offsets of lengths drawn like adapter / quality-trimmed reads: 70 % untouched (150), the rest
    # uniform on 30..149.  Checked in closed form against the lengths themselves.
    ragged = None
    if be.name == "hip" and not a.full_matrix and n >= 1000 and not a.no_ragged and L > 30:
        g0 = torch.Generator(device="cuda").manual_seed(2025 + rank)
        rl = torch.randint(30, L, (n,), device="cuda", generator=g0, dtype=torch.int64)   # 30 .. L-1: never beyond a read's bytes
        keep = torch.rand(n, device="cuda", generator=g0) < 0.7
        rl[keep] = L
        del keep
        ro = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(rl, 0, out=ro[1:])
        tot = int(ro[-1].item())
        assert tot <= d_qual.numel(), "ragged offsets reach beyond the resident quality bytes"
        want_hist = torch.bincount(rl, minlength=512).cpu().numpy()
        del rl
        r_ms = []
        for i in range(4):
            ctx.fastq_tally_dev(d_qual, ro, n, flags=0)
            got = ctx.fastq_tally_fetch()
            if i:
                r_ms.append(ctx.last_kernel_ms(0))
        assert got.total == tot and (got.seqlen == want_hist).all(), "K1 on ragged reads: SeqLen / sum differ from the lengths drawn"
        r_k = sorted(r_ms)[len(r_ms) // 2]
        r_bytes = tot + (n + 1) * 8
        ragged = {"workload": f"the same {n:.3g} reads cut to mixed lengths (70 % {L}, 30 % uniform 30..{L - 1}: {tot / n:.1f} bp on average)",
                  "kernel_ms": round(r_k, 4), "algorithmic_bytes_per_launch": r_bytes, "achieved": round(r_bytes / r_k / 1e6, 1),
                  "frac": round(r_bytes / r_k / 1e6 / HBM_PEAK_GBS, 4), "exact": "SeqLen[512] and sum equal the lengths drawn"}
        del ro
        torch.cuda.empty_cache()

    # ---- result checks (after the timed region) ------------------------------------------------
    # closed form: every record of every rank counted once, at its length
    job.check_closed_form(out, n, L)
    extra = {}
    if not a.no_extra:
        # exact: this rank's K1 counts against an independent kernel (K1L) over the same resident bytes, and three
        # 2e5-read windows against the CPU oracle (bench_extra.exact_check)
        import bench_extra
        ctx.fastq_tally_dev(d_qual, d_off, n, flags=0)
        loc = ctx.fastq_tally_fetch()
        local = {"seqlen": loc.seqlen.copy(), "total": loc.total, "q20": loc.q20, "q30": loc.q30}
        extra["exact"] = bench_extra.exact_check(ctx, d_qual, d_off, n, L, 12345, first, local)
        if world == 1:
            assert (out["total"], out["q20"], out["q30"]) == (local["total"], local["q20"], local["q30"])
    del d_qual, d_off
    if be.name == "hip":
        torch.cuda.empty_cache()
    if world == 1 and not a.no_extra:
        extra["plain_traffic"] = bench_extra.plain_traffic()
        extra["kernel_legs"] = bench_extra.kernel_legs(ctx)
        try:
            extra["end_to_end"] = bench_extra.e2e_legs(ctx, usable_cpus())
        except Exception as e:  # noqa: BLE001  (file-system or toolchain trouble must not take the GPU line down)
            extra["end_to_end"] = [{"leg": "failed", "why": str(e)[:300]}]

    if rank == 0:
        bases = world * n * L * a.steps
        alg_bytes = n * L + (n + 1) * 8  # 1 B per base + 8 B per record, per launch
        k_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: rocprofv3 cannot run inside this process, so this is the figure of the
        # separate `--pmc` passes over this same command, kept in profiles/traffic.json (traffic_source says which run)
        traffic, traffic_source = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                t = json.load(open(tj))
                if t.get("reads_per_launch") == n and t.get("read_len") == L and not a.full_matrix:
                    traffic = t.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, round {t.get('round')}; not measured by this run)"
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "Gbases/s processed (fastq_count, 150 bp)", "value": round(bases / dt / 1e9, 3),
            "unit": "Gbases/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"fastq_count on {n:.3g} x {L} bp synthetic reads per GPU, resident in HBM "
                                   f"(BASELINE configs[1]; inflate and PCIe are outside the timed region: extra.end_to_end has them)",
                       "reads_per_gpu": n, "read_len": L, "kernel": "k_tally_hist" if a.full_matrix else "k_tally_scan",
                       "outputs": "SeqLen[512], sum, Q20, Q30" + (", Quality[128][512]" if a.full_matrix else ""),
                       "parallelism": f"record-block shard x{world}", "allreduce": allreduce,
                       "rccl_library": rccl_lib, "rccl_ranks": rccl_ranks, "ranks": world,
                       "launcher": os.environ.get("HPN_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else ("none" if world == 1 else "external")),
                       **({} if be.name == "hip" else {"backend": be.name})},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(k_ms, 4), "kernel_ms_per_rank": [round(x, 4) for x in k_ranks],
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "frac_ragged": ragged["frac"] if ragged else None, "ragged": ragged},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(L, a.cpu_seconds)
            except Exception as e:  # noqa: BLE001  (a reported baseline must not take the GPU line down)
                line["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": usable_cpus(), "kind": "port",
                                        "sample": f"failed: {e}"}
        if extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
