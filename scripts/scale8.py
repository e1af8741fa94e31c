#!/usr/bin/env python3
"""ONE command for the first multi-GPU box (BASELINE configs[4]): everything the one-GPU boxes could not
measure, into one JSON object.

    python scripts/scale8.py [shm_dir] [out.json]            (scripts/scale8.sh wraps it and copies the JSON to profiles/)
    SCALE8_GB=64      text size of the plain and of the gzip FASTQ file (GB; default 64)
    SCALE8_DEPTH=30   depth of the C4 BAM (all 25 hg38 contigs; 30 = the stated size, 93 GB of BAM)
    SCALE8_GPUS=1,2,4,8   the rank counts of the bench leg (those the box has)

What it runs, each leg with outputs compared:
  1. `bench.py --gpus N --no-extra` for N = 1, 2, 4, 8: the weak-scaling lines (1e9 reads per rank, ONE RCCL sum per step), with
     `config.rccl_ranks` -- a rank whose collective does not span N ranks prints no line.
  2. `fastq_count` on one plain and one gzip FASTQ file of SCALE8_GB of text: with the default choice of lanes (`lanes_worth`,
     host/cpus.hpp) and with HPN_NGPU = every device; rows compared with each other and with the closed form; the tools'
     `[hpn] N lanes on devices ...; counts summed by RCCL (N ranks)` lines captured.
  3. `bam2depth` and `bam_sliding_count` on the C4 file: default workers and HPN_NGPU = every device; every output file compared
     byte for byte between the two (and the one-device run of a one-device box is scripts/c4_full.py's, against the oracle).
  4. The C tools' collective library: `tests/abi/comm_one_rank.c` built and run (no torch in the process): the path of the
     librccl that carried the sum.

Reference seams: reduceStats fastq_count_kthread.c:180-210 (what the RCCL sum stands in for); kt_for over files
fastq_count_kthread.c:270; the per-target loop bam2depth.c:325-339.  Nothing here imports oracle/."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
LIBDIR = os.path.join(ROOT, "highperformancengs_amd")


def n_devices():
    # (device_count() alone does not initialise the runtime; still a child process: this one stays free of it)
    p = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        return int(p.stdout.decode().strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def hpn_lines(err):
    return [l for l in err.decode(errors="replace").splitlines() if l.startswith("[hpn]")]


def timed(cmd, cwd, env=None):
    time.sleep(1.5)                       # (a GPU process that starts right behind another one's exit pays for its teardown: DESIGN 5.1)
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1", **(env or {})})
    return time.perf_counter() - t0, p


def file_md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def bench_leg(gpus, steps=20, warmup=3):
    out = []
    for n in gpus:
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup), "--no-extra"],
                           cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        line = None
        for l in p.stdout.decode().splitlines():
            if l.startswith("{"):
                line = json.loads(l)
        out.append({"gpus": n, "rc": p.returncode, "wall_s": round(time.perf_counter() - t0, 1), "line": line,
                    "stderr_tail": p.stderr.decode(errors="replace").splitlines()[-4:] if p.returncode or line is None else []})
    one = next((o["line"]["value"] for o in out if o["gpus"] == 1 and o["line"]), None)
    for o in out:
        if one and o["line"]:
            o["value_over_n_times_one_gpu"] = round(o["line"]["value"] / (o["gpus"] * one), 4)     # (the driver computes its own; this is for the reader)
    return out


def fastq_legs(td, gb, ndev):
    import torch  # noqa: F401
    import highperformancengs_amd as hp
    import bench_extra
    per, L, K = 100_000, 150, 50
    ctx = hp.Context(0)
    texts = [bench_extra._fastq_text(ctx, per, L, 100 + k).tobytes() for k in range(K)]
    ctx.close()
    cycle_text = b"".join(texts)
    cycles = max(1, int(gb * 1e9 / len(cycle_text)))
    with ThreadPoolExecutor(16) as ex:
        cycle_gz = b"".join(ex.map(bench_extra._gzip_one, texts))
    del texts
    legs = []
    for name, blob in (("plain.fq", cycle_text), ("members.fq.gz", cycle_gz)):
        path = os.path.join(td, name)
        fd = os.open(path, os.O_CREAT | os.O_WRONLY, 0o644)
        os.ftruncate(fd, len(blob) * cycles)
        with ThreadPoolExecutor(16) as ex:
            list(ex.map(lambda c: os.pwrite(fd, blob, c * len(blob)), range(cycles)))
        os.close(fd)
        one = os.path.join(td, "cycle_" + name)
        open(one, "wb").write(blob)
        row1 = subprocess.run([os.path.join(BIN, "fastq_count"), os.path.basename(one)], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout.decode().strip().splitlines()[-1].split("\t")
        want = [name, str(int(row1[1]) * cycles), str(int(row1[2]) * cycles)] + row1[3:]
        runs = []
        for label, env in (("default lanes (lanes_worth)", {}), (f"HPN_NGPU={ndev}", {"HPN_NGPU": str(ndev)}), ("HPN_NGPU=1", {"HPN_NGPU": "1"})):
            best = None
            for _ in range(2):
                dt, p = timed([os.path.join(BIN, "fastq_count"), name], td, env)
                row = p.stdout.decode().strip().splitlines()[-1].split("\t") if p.stdout.strip() else []
                r = {"run": label, "seconds": round(dt, 3), "rc": p.returncode, "gbases_per_s": round(cycles * K * per * L / dt / 1e9, 2),
                     "input_GBps": round(os.path.getsize(path) / dt / 1e9, 2), "row_identical": row == want,
                     "said": [l for l in hpn_lines(p.stderr) if "lanes" in l or "summed" in l or "workers" in l][:3]}
                if best is None or r["seconds"] < best["seconds"]:
                    best = r
            runs.append(best)
        legs.append({"input": f"{name}: {cycles * K * per:.3e} x {L} bp, {os.path.getsize(path) / 1e9:.1f} GB on disk", "expected_row": want, "runs": runs})
        os.unlink(path)
    return legs


def bam_legs(td, disk, depth, ndev, threads=15):
    import c4
    tg = c4.targets(lambda n, l: depth)
    n_reads = sum(r for _, _, r in tg)
    if shutil.disk_usage(td).free < n_reads * 175 * 1.3 or shutil.disk_usage(disk).free < n_reads * 92:
        return {"skipped": f"needs {(n_reads * 175 * 13 // 10) >> 30} GiB in {td} and {(n_reads * 92) >> 30} GiB in {disk}"}
    sd = tempfile.mkdtemp(prefix="scale8_soa_", dir=disk)
    exe = c4.build_synth(sd)
    bam = os.path.join(td, "hg38.bam")
    subprocess.check_call([exe, bam, "--targets", ",".join(f"{n}:{l}:{r}" for n, l, r in tg), str(threads), os.path.join(sd, "hg38.soa")])
    shutil.rmtree(sd, ignore_errors=True)
    out = {"input": f"{n_reads:.3e} x 150 bp over the 25 hg38 contigs at {depth:g}x; BAM {os.path.getsize(bam) / 1e9:.1f} GB", "runs": []}
    for tool, args, files in (("bam2depth", ["-w", "20000", "-o", "d", "hg38.bam"], ["hg38.bam.1.bedGraph", "d.1.depth"]),
                              ("bam_sliding_count", ["-w", "20000", "-o", "s", "hg38.bam"], ["s.txt"])):
        sums = {}
        for label, env in (("HPN_NGPU=1", {"HPN_NGPU": "1"}), ("default workers (lanes_worth)", {}), (f"HPN_NGPU={ndev}", {"HPN_NGPU": str(ndev)})):
            wd = tempfile.mkdtemp(prefix="run_", dir=td)
            os.symlink(bam, os.path.join(wd, "hg38.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38.bam.bai"))
            dt, p = timed([os.path.join(BIN, tool)] + args, wd, env)
            md5 = {f: file_md5(os.path.join(wd, f)) if os.path.exists(os.path.join(wd, f)) else None for f in files}
            sums[label] = md5
            out["runs"].append({"run": f"{tool} {label}", "seconds": round(dt, 3), "rc": p.returncode, "gbases_per_s": round(n_reads * 150 / dt / 1e9, 2),
                                "md5": md5, "identical_to_one_device": md5 == sums["HPN_NGPU=1"] and None not in md5.values(),
                                "said": [l for l in hpn_lines(p.stderr) if "workers" in l or "lanes" in l or "devices" in l][:3]})
            shutil.rmtree(wd, ignore_errors=True)
    os.unlink(bam)
    return out


def rccl_of_the_c_tools():
    exe = os.path.join(tempfile.mkdtemp(prefix="scale8_exe_"), "comm_one_rank")       # (not under /dev/shm: usually mounted noexec)
    cc = subprocess.run(["gcc", "-std=c99", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi", "comm_one_rank.c"), "-o", exe,
                         "-L" + LIBDIR, "-lhpngs", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if cc.returncode:
        return {"built": False, "why": cc.stdout.decode()[-300:]}
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return {"built": True, "rc": p.returncode, "librccl": (p.stdout.decode().strip().splitlines() or [""])[-1], "stderr": p.stderr.decode()[-300:]}


def main():
    shm = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
    dest = sys.argv[2] if len(sys.argv) > 2 else None
    ndev = n_devices()
    gpus = [int(x) for x in os.environ.get("SCALE8_GPUS", "1,2,4,8").split(",") if int(x) <= max(ndev, 1)]
    out = {"devices": ndev, "note": None if ndev >= 8 else f"this box has {ndev} device(s): the legs run with what is there and say so"}
    td = tempfile.mkdtemp(prefix="scale8_", dir=shm)
    try:
        out["rccl_of_the_c_tools"] = rccl_of_the_c_tools()
        out["bench"] = bench_leg(gpus)
        out["fastq_count"] = fastq_legs(td, float(os.environ.get("SCALE8_GB", "64")), max(ndev, 1))
        out["bam"] = bam_legs(td, tempfile.gettempdir(), float(os.environ.get("SCALE8_DEPTH", "30")), max(ndev, 1))
    finally:
        shutil.rmtree(td, ignore_errors=True)
    s = json.dumps(out)
    print(s)
    if dest:
        open(dest, "w").write(s + "\n")


if __name__ == "__main__":
    main()
