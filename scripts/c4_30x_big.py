#!/usr/bin/env python3
"""BASELINE configs[3] from a FILE with the big contigs at full depth: 30x on chr1 .. chr7 (~2.5e8 reads, ~38 GB of BAM -- the whole
genome at 30x is ~90 GB, more than the GPU box's 79 GB disk), 3x on the other 18 contigs.  bam2depth runs on one worker and with the
targets over three workers (HPN_NGPU=3: the look-ahead budget of host/bam_multi.hpp under load); every byte of the bedGraph and of
the depth report is compared with the oracle run on the generator's own records, target by target (bam2depth.c:325-339 is the loop);
bam_sliding_count's report on three workers against one.  Peak host RSS of every run is reported (the 12 GB look-ahead budget).

    python scripts/c4_30x_big.py [workdir] [threads]      -> one JSON object on stdout; copy it to profiles/r04/c4_30x_big.json
    C4_PACKED=1: the same records written htsjdk's way (packed across BGZF blocks)   -> profiles/r04/c4_30x_big_packed.json

Checker use of oracle/ (tests/c4.py); the product runs are the built binaries."""
import json
import os
import resource
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
BIG = {"chr1", "chr2", "chr3", "chr4", "chr5", "chr6", "chr7"}


def run(tool, args, wd, env):
    """-> (seconds, rc, peak RSS in MB of the child, stderr tail)"""
    t0 = time.perf_counter()
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    p = subprocess.run([os.path.join(BIN, tool)] + args, cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1", **env})
    dt = time.perf_counter() - t0
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss      # (high-water mark over all children so far: runs are ordered small -> large)
    look = [l for l in p.stderr.decode().splitlines() if "look-ahead" in l]
    return dt, p.returncode, max(rss, before) / 1024.0, look[-1] if look else ""


def main():
    td = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="c4big_")
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else max(2, (os.cpu_count() or 4) - 1)
    os.makedirs(td, exist_ok=True)
    if shutil.disk_usage(td).free < (75 << 30):
        print(json.dumps({"skipped": f"{shutil.disk_usage(td).free >> 30} GiB free in {td}: 75 needed"}))
        return
    tg = c4.targets(lambda n, l: 30.0 if n in BIG else 3.0)
    n_reads = sum(r for _, _, r in tg)
    t0 = time.perf_counter()
    packed = os.environ.get("C4_PACKED") == "1"          # htsjdk's layout: records packed across BGZF blocks (bam_synth's BAM_SYNTH_PACKED)
    bam, prefix = c4.synth(td, "hg38_big.bam", tg, threads, env={"BAM_SYNTH_PACKED": "1"} if packed else None)
    t_synth = time.perf_counter() - t0
    os.unlink(prefix + ".seq4")                      # 19 GB the depth check does not need
    open(prefix + ".seq4", "wb").close()
    soa = c4.Soa.__new__(c4.Soa)                     # the record arrays without the sequences
    import numpy as np
    soa.tid, soa.pos = np.fromfile(prefix + ".tid", np.int32), np.fromfile(prefix + ".pos", np.int32)
    soa.flag, soa.kind = np.fromfile(prefix + ".flag", np.uint32), np.fromfile(prefix + ".kind", np.uint8)
    soa.n = len(soa.tid)
    ncig = np.array([len(k) for k in c4.KIND_CIGAR], np.uint32)[soa.kind]
    soa.cigar_off = np.zeros(soa.n + 1, np.uint32)
    np.cumsum(ncig, out=soa.cigar_off[1:])
    table = np.zeros((4, 3), np.uint32)
    for k, ops in enumerate(c4.KIND_CIGAR):
        table[k, :len(ops)] = ops
    soa.cigar = np.concatenate([table[soa.kind[a:a + (1 << 24)]][np.arange(3)[None, :] < ncig[a:a + (1 << 24), None]] for a in range(0, soa.n, 1 << 24)]).astype(np.uint32)
    soa.lo = np.searchsorted(soa.tid, np.arange(len(tg) + 1))
    for ext in (".tid", ".pos", ".flag", ".kind", ".seq4"):
        os.unlink(prefix + ext)
    W = 20000
    out = {"input": f"{n_reads:.3e} x 150 bp over the 25 hg38 contigs: 30x on chr1-chr7, 3x elsewhere; BAM {os.path.getsize(bam) / 1e9:.1f} GB"
                    + (", records packed across BGZF blocks" if packed else ""),
           "input_made_in_s": round(t_synth, 1), "runs": []}
    per_target = []
    for label, env in (("bam2depth, one worker", {}), ("bam2depth, targets over three workers (HPN_NGPU=3)", {"HPN_NGPU": "3"})):
        wd = tempfile.mkdtemp(prefix="run_", dir=td)
        os.symlink(bam, os.path.join(wd, "hg38_big.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38_big.bam.bai"))
        dt, rc, rss, look = run("bam2depth", ["-w", str(W), "-o", "d", "hg38_big.bam"], wd, env)
        ok, n_runs, cov_sum = rc == 0, 0, 0
        with open(os.path.join(wd, "hg38_big.bam.1.bedGraph"), "rb") as fb, open(os.path.join(wd, "d.1.depth"), "rb") as fd:
            for t in range(len(tg)):
                runs, bins = c4.oracle_depth_target(soa, tg, t, W)
                bed, dep = c4.oracle_target_text(tg[t][0], tg[t][1], W, runs, bins)
                same = fb.read(len(bed)) == bed and fd.read(len(dep)) == dep
                ok = ok and same
                n_runs += len(runs)
                if not per_target or len(per_target) < len(tg):
                    per_target.append({"target": tg[t][0], "runs": int(len(runs)), "sum_len_x_depth": int(((runs[:, 1] - runs[:, 0]).astype("int64") * runs[:, 2]).sum()) if len(runs) else 0,
                                       "window_sum": float(bins.sum()), "identical": bool(same)})
                del runs, bins, bed, dep
            ok = ok and fb.read(1) == b"" and fd.read(1) == b""
        out["runs"].append({"run": label, "seconds": round(dt, 2), "gbases_per_s": round(n_reads * 150 / dt / 1e9, 2), "rc": rc, "peak_child_rss_MB": round(rss, 1),
                            "outputs_identical_to_oracle": bool(ok), "bedgraph_lines": n_runs, "look_ahead": look})
        shutil.rmtree(wd, ignore_errors=True)
    out["targets"] = per_target
    rep = {}
    for label, env in (("bam_sliding_count, one worker (HPN_NGPU=1)", {"HPN_NGPU": "1"}), ("bam_sliding_count, three workers", {"HPN_NGPU": "3"})):
        wd = tempfile.mkdtemp(prefix="run_", dir=td)
        os.symlink(bam, os.path.join(wd, "hg38_big.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38_big.bam.bai"))
        dt, rc, rss, _ = run("bam_sliding_count", ["-w", str(W), "-o", "s", "hg38_big.bam"], wd, env)
        rep[label] = open(os.path.join(wd, "s.txt"), "rb").read() if rc == 0 else None
        out["runs"].append({"run": label, "seconds": round(dt, 2), "gbases_per_s": round(n_reads * 150 / dt / 1e9, 2), "rc": rc, "peak_child_rss_MB": round(rss, 1)})
        shutil.rmtree(wd, ignore_errors=True)
    vals = list(rep.values())
    out["sliding_reports_identical_one_vs_three_workers"] = bool(vals[0] is not None and vals[0] == vals[1])
    print(json.dumps(out))
    shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
