#!/usr/bin/env python3
"""BASELINE configs[3] AT ITS STATED SIZE, from a file: all 25 hg38 primary contigs at 30x (~6.2e8 reads x 150 bp, ~93 GB of BAM --
more than the GPU box's 79 GB disk, so the BAM lives in /dev/shm; the generator's record arrays go to the disk and are loaded for
the oracle).  bam2depth and bam_sliding_count run once each on their defaults (one worker on one device); EVERY byte of the
bedGraph, of the depth report and of bam_sliding_count's report is compared with the oracle run on the generator's own records
(per target: bam2depth.c:325-339 is the loop; bam_sliding_count.c:389-416), wall time and peak RSS are taken, and both tools run
once more under rocprofv3 --kernel-trace --stats for the per-kernel totals.

    python scripts/c4_full.py [shm_dir] [disk_dir] [threads]      -> one JSON object on stdout; copy it to profiles/r05/c4_full.json
    C4_DEPTH=x: another depth (a dry run of the script at 3x takes a minute)

Checker use of oracle/ (tests/c4.py); the product runs are the built binaries."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import c4  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")


def run(tool, args, wd, env=None, prof=None):
    cmd = [os.path.join(BIN, tool)] + args
    e = {**os.environ, "HPN_TIMING": "1", **(env or {})}
    if prof:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", prof, "-o", "t", "--"] + cmd
        e["HPN_FULL_EXIT"] = "1"
        e["TMPDIR"] = "/tmp"
    # peak RSS: VmHWM of the child's own address space, polled while it runs (ru_maxrss of a child forked from a process that
    # holds 70 GB of arrays starts at the PARENT's size: exec records the old mm's high-water mark)
    import threading
    hwm = [0.0]

    def poll(pid):
        while True:
            try:
                for line in open(f"/proc/{pid}/status"):
                    if line.startswith("VmHWM:"):
                        hwm[0] = max(hwm[0], int(line.split()[1]) / 1024.0)
            except OSError:
                return
            time.sleep(0.05)
    t0 = time.perf_counter()
    with open(os.path.join(wd, "stderr.txt"), "wb") as fe:
        p = subprocess.Popen(cmd, cwd=wd, stdout=subprocess.DEVNULL, stderr=fe, env=e)
        th = threading.Thread(target=poll, args=(p.pid,), daemon=True)
        th.start()
        p.wait()
    dt = time.perf_counter() - t0
    return dt, p.returncode, hwm[0], open(os.path.join(wd, "stderr.txt"), errors="replace").read()


def kernel_totals(prof):
    f = glob.glob(os.path.join(prof, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return None
    rows = list(csv.DictReader(open(f[0])))
    out = [{"kernel": r["Name"].split("(")[0].replace("void ", "").replace("hpn::", ""), "calls": int(r["Calls"]),
            "total_ms": round(int(r["TotalDurationNs"]) / 1e6, 3), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4)} for r in rows[:8]]
    return {"device_ms": round(sum(int(r["TotalDurationNs"]) for r in rows) / 1e6, 1), "top": out}


def main():
    import numpy as np
    shm = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
    disk = sys.argv[2] if len(sys.argv) > 2 else tempfile.gettempdir()
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    depth = float(os.environ.get("C4_DEPTH", "30"))
    td = tempfile.mkdtemp(prefix="c4full_", dir=shm)
    sd = tempfile.mkdtemp(prefix="c4full_soa_", dir=disk)
    tg = c4.targets(lambda n, l: depth)
    n_reads = sum(r for _, _, r in tg)
    need_shm, need_disk = n_reads * 175, n_reads * 92
    if shutil.disk_usage(td).free < need_shm or shutil.disk_usage(sd).free < need_disk:
        print(json.dumps({"skipped": f"needs {need_shm >> 30} GiB in {shm} and {need_disk >> 30} GiB in {disk}"}))
        return
    t0 = time.perf_counter()
    exe = c4.build_synth(sd)
    bam = os.path.join(td, "hg38_30x.bam")
    prefix = os.path.join(sd, "hg38_30x.soa")
    subprocess.check_call([exe, bam, "--targets", ",".join(f"{n}:{l}:{r}" for n, l, r in tg), str(threads), prefix])
    t_synth = time.perf_counter() - t0
    soa = c4.Soa(prefix, len(tg))
    shutil.rmtree(sd, ignore_errors=True)
    W = 20000
    out = {"input": f"{n_reads:.3e} x 150 bp over the 25 hg38 contigs at {depth:g}x; BAM {os.path.getsize(bam) / 1e9:.1f} GB in {shm}",
           "input_made_in_s": round(t_synth, 1), "runs": []}
    # ---- bam2depth ----
    wd = tempfile.mkdtemp(prefix="run_", dir=td)
    os.symlink(bam, os.path.join(wd, "hg38_30x.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38_30x.bam.bai"))
    dt1, rc1, rss1, _ = run("bam2depth", ["-w", str(W), "-o", "d", "hg38_30x.bam"], wd)
    for f in ("hg38_30x.bam.1.bedGraph", "d.1.depth"):
        os.unlink(os.path.join(wd, f))
    dt, rc, rss, err = run("bam2depth", ["-w", str(W), "-o", "d", "hg38_30x.bam"], wd)
    ok, n_runs, per_target = rc == 0 and rc1 == 0, 0, []
    t_or = time.perf_counter()
    with open(os.path.join(wd, "hg38_30x.bam.1.bedGraph"), "rb") as fb, open(os.path.join(wd, "d.1.depth"), "rb") as fd:
        for t in range(len(tg)):
            runs, bins = c4.oracle_depth_target(soa, tg, t, W)
            bed, dep = c4.oracle_target_text(tg[t][0], tg[t][1], W, runs, bins)
            same = fb.read(len(bed)) == bed and fd.read(len(dep)) == dep
            ok = ok and same
            n_runs += len(runs)
            per_target.append({"target": tg[t][0], "runs": int(len(runs)), "identical": bool(same)})
            del runs, bins, bed, dep
        ok = ok and fb.read(1) == b"" and fd.read(1) == b""
    out["runs"].append({"run": "bam2depth -w 20000 (default: one worker), second of two runs", "seconds": round(dt, 2), "first_run_seconds": round(dt1, 2),
                        "gbases_per_s": round(n_reads * 150 / dt / 1e9, 2), "rc": rc,
                        "peak_rss_MB": round(rss, 1), "outputs_identical": bool(ok), "compared_with": "oracle (orc_depth_target per target), every byte",
                        "bedgraph_lines": n_runs, "bedgraph_GB": round(os.path.getsize(os.path.join(wd, "hg38_30x.bam.1.bedGraph")) / 1e9, 2),
                        "oracle_s": round(time.perf_counter() - t_or, 1), "stderr_tail": [l for l in err.splitlines() if l.startswith("[hpn]")][-3:]})
    out["targets"] = per_target
    shutil.rmtree(wd, ignore_errors=True)
    # ---- bam_sliding_count ----
    wd = tempfile.mkdtemp(prefix="run_", dir=td)
    os.symlink(bam, os.path.join(wd, "hg38_30x.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38_30x.bam.bai"))
    dt1, rc1, _, _ = run("bam_sliding_count", ["-w", str(W), "-o", "s", "hg38_30x.bam"], wd)
    os.unlink(os.path.join(wd, "s.txt"))
    dt, rc, rss, err = run("bam_sliding_count", ["-w", str(W), "-o", "s", "hg38_30x.bam"], wd)
    t_or = time.perf_counter()
    got = open(os.path.join(wd, "s.txt"), "rb").read() if rc == 0 else b""
    want = c4.oracle_window_report(soa, tg, W)
    out["runs"].append({"run": "bam_sliding_count -w 20000 (default: one worker), second of two runs", "seconds": round(dt, 2), "first_run_seconds": round(dt1, 2),
                        "gbases_per_s": round(n_reads * 150 / dt / 1e9, 2), "rc": rc,
                        "peak_rss_MB": round(rss, 1), "outputs_identical": bool(rc == 0 and got == want),
                        "compared_with": "oracle (orc_window_add + float32 replay), every byte", "report_bytes": len(got), "oracle_s": round(time.perf_counter() - t_or, 1)})
    shutil.rmtree(wd, ignore_errors=True)
    del soa, want
    out["outputs_identical"] = all(r["outputs_identical"] for r in out["runs"])
    # ---- the same two runs under rocprofv3 (per-kernel totals; their wall times are the profiler's, not the tools') ----
    for tool, args in (("bam2depth", ["-w", str(W), "-o", "d", "hg38_30x.bam"]), ("bam_sliding_count", ["-w", str(W), "-o", "s", "hg38_30x.bam"])):
        wd = tempfile.mkdtemp(prefix="prof_", dir=td)
        os.symlink(bam, os.path.join(wd, "hg38_30x.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38_30x.bam.bai"))
        prof = os.path.join(wd, "prof")
        dt, rc, _, _ = run(tool, args, wd, prof=prof)
        out["runs"].append({"run": f"{tool} under rocprofv3 --kernel-trace --stats", "seconds": round(dt, 2), "rc": rc, "kernels": kernel_totals(prof)})
        shutil.rmtree(wd, ignore_errors=True)
    print(json.dumps(out))
    shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
