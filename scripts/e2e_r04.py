"""Round 4's end-to-end A/B runs on the GPU box, with the tools' own timing lines (HPN_TIMING):
  * fastq_count on a 7.2 GB .fastq.gz of three members (the bench's gzip leg): one context / two and three lanes on the one device
    (host/gz_shard.hpp: the lanes share the chip here; on a node every lane has its own);
  * bam2depth / bam_sliding_count on the C4-shaped 25-contig BAM (10 GB): the compressed bytes of the next launch copied beside
    the kernels (HPN_BAM_AHEAD, default) or in line with them (=0).
   python scripts/e2e_r04.py [gz|bam|all] > gpurun_out/e2e_r04.txt"""
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

BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
td = tempfile.mkdtemp(prefix="e2e_r04_")


def run(tool, args, env, reps=2, cwd=td):
    best, err, out = 1e9, "", b""
    for _ in range(reps):
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(BIN, tool)] + args, cwd=cwd, env={**os.environ, "HPN_TIMING": "1", **env}, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        if dt < best:
            best, err, out = dt, p.stderr.decode(), p.stdout
    lines = [l for l in err.splitlines() if (l.startswith("[hpn") or l.startswith("Finished")) and "context 0.0" not in l]
    print(f"--- {tool} {' '.join(args)}  {env}: {best:.3f} s")
    for l in lines[-(60 if env.get("HPN_TIMING") == "2" else 6):]:
        print("    " + l[:260])
    sys.stdout.flush()
    return out


if what in ("gz", "all"):
    n = 13_000_000
    ctx = hp.Context(0)
    raw = bench_extra._fastq_text(ctx, n, 150, 40).tobytes()
    ctx.close()
    one = bench_extra._gz_single_member(raw, 256, 16)
    with open(os.path.join(td, "gz3.fq.gz"), "wb") as f:
        for _ in range(3):
            f.write(one)
    print(f"gz3.fq.gz: {3 * len(one) / 1e9:.2f} GB compressed, {3 * len(raw) / 1e9:.2f} GB of text, {3 * n} reads")
    del raw, one
    outs = [run("fastq_count", ["gz3.fq.gz"], e) for e in ({}, {"HPN_GZ_OVERLAP": "1"}, {"HPN_TIMING": "2"}, {"HPN_NGPU": "2"}, {"HPN_NGPU": "3"}, {"HPN_GZ_FIND": "device"}, {"HPN_GZ_FIND": "device", "HPN_TIMING": "2"}, {"HPN_NGPU": "2", "HPN_GZ_FIND": "device"})]
    print("outputs identical:", all(o == outs[0] for o in outs), outs[0].decode().strip())
    os.unlink(os.path.join(td, "gz3.fq.gz"))

if what in ("bam", "all"):
    import hashlib
    import c4
    tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
    seen = {}
    # the same records twice: samtools' layout (no record crosses a BGZF block) and htsjdk's (records packed across blocks,
    # BAM_SYNTH_PACKED=1; .bai offsets into the middle of blocks) -- the second is decoded on the device since round 4
    for label, env in (("record-aligned blocks", None), ("records packed across blocks", {"BAM_SYNTH_PACKED": "1"})):
        sub = tempfile.mkdtemp(dir=td)
        bam, prefix = c4.synth(sub, "hg38.bam", tg, 15, soa=False, env=env)
        print(f"hg38.bam ({label}): {os.path.getsize(bam) / 1e9:.1f} GB, {sum(r for _, _, r in tg)} reads")
        for tool, args in (("bam2depth", ["-w", "20000", "-o", "d", "hg38.bam"]), ("bam_sliding_count", ["-w", "20000", "-o", "s", "hg38.bam"])):
            res = {}
            for e in ({}, {"HPN_BAM_AHEAD": "0"}, {"HPN_NGPU": "3"} if tool == "bam2depth" else {"HPN_NGPU": "1"}):
                wd = tempfile.mkdtemp(dir=td)
                os.symlink(bam, os.path.join(wd, "hg38.bam")), os.symlink(bam + ".bai", os.path.join(wd, "hg38.bam.bai"))
                run(tool, args, e, cwd=wd)
                outs = sorted(f for f in os.listdir(wd) if not f.startswith("hg38.bam") or f.endswith("bedGraph"))
                res[str(e)] = [(f, hashlib.md5(open(os.path.join(wd, f), "rb").read()).hexdigest()) for f in outs if os.path.isfile(os.path.join(wd, f)) and not os.path.islink(os.path.join(wd, f))]
                subprocess.run(["rm", "-rf", wd])
            vals = list(res.values())
            print(f"{tool}: outputs identical across routes:", all(v == vals[0] for v in vals), [x[0] for x in vals[0]])
            if tool in seen:
                print(f"{tool}: outputs identical to the record-aligned file's:", vals[0] == seen[tool])
            seen.setdefault(tool, vals[0])
        subprocess.run(["rm", "-rf", sub])
if what == "rounds":      # chunks of 88 MB under one inflate launch (HPN_BAM_ROUNDS): 6,144 decoder waves take ~1.4 chunks at once
    import c4
    tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
    bam, prefix = c4.synth(td, "hg38.bam", tg, 15, soa=False)
    for r in ("3", "4", "6", "8", "12"):
        for tool, args in (("bam2depth", ["-w", "20000", "-o", "d", "hg38.bam"]), ("bam_sliding_count", ["-w", "20000", "-o", "s", "hg38.bam"])):
            run(tool, args, {"HPN_BAM_ROUNDS": r, "HPN_NGPU": "1"}, reps=3)
subprocess.run(["rm", "-rf", td])
