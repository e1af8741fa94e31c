#!/usr/bin/env python3
"""BASELINE configs[1] IN ITS STATED INPUT FORM: fastq_count on 1e9 x 150 bp reads as ONE gzip file (multi-member, one member
per 1e5 records, zlib level 1: SURVEY 8d).  The file -- ~155 GB, ~310 GB of text -- is more than the box's disk: it is laid out in
/dev/shm from 50 DISTINCT members (seeds 100 .. 149 of the counter-based generator, 1e5 reads each) written 200 times over
(SURVEY 8d allows a logical 1e9 reads to be realised by re-submitting a batch; the tool sees one ordinary file of 10,000 members and
knows nothing of the repetition).  Expected row: the REFERENCE binary's own row (oracle/_ref/fastq_count; the oracle restatement
where that is absent) for one cycle of the 50 members, with ReadCount and BaseCount x 200 -- mean, min, max and the Q20 / Q30
percentages of 200 identical cycles are the cycle's.

    python scripts/c2_gz_1e9.py [shm_dir] [cycles]     -> one JSON object on stdout; copy it to profiles/r05/c2_gz_1e9.json

Reference loop: the four gzgets of count_read, fastq_count.c:112-119."""
import json
import os
import subprocess
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import highperformancengs_amd as hp  # noqa: E402
import bench_extra  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "fastq_count")


def main():
    shm = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    K, per, L = 50, 100_000, 150
    td = tempfile.mkdtemp(prefix="c2gz_", dir=shm)
    ctx = hp.Context(0)
    t0 = time.perf_counter()
    texts = [bench_extra._fastq_text(ctx, per, L, 100 + k).tobytes() for k in range(K)]
    ctx.close()
    with ThreadPoolExecutor(16) as ex:
        members = list(ex.map(bench_extra._gzip_one, texts))
    cycle = b"".join(members)
    text_bytes = sum(len(t) for t in texts)
    del texts
    one = os.path.join(td, "cycle.fq.gz")
    open(one, "wb").write(cycle)
    big = os.path.join(td, "c2.fq.gz")
    fd = os.open(big, os.O_CREAT | os.O_WRONLY, 0o644)
    os.ftruncate(fd, len(cycle) * cycles)
    with ThreadPoolExecutor(16) as ex:      # (os.pwrite releases the GIL)
        list(ex.map(lambda c: os.pwrite(fd, cycle, c * len(cycle)), range(cycles)))
    os.close(fd)
    t_made = time.perf_counter() - t0
    out = {"input": f"{cycles * K * per:.3e} x {L} bp as ONE gzip file of {cycles * K} members ({K} distinct, written {cycles} times): "
                    f"{len(cycle) * cycles / 1e9:.1f} GB compressed, {text_bytes * cycles / 1e9:.1f} GB of text, in {shm}", "input_made_in_s": round(t_made, 1)}
    # ---- one cycle through the reference (or the oracle): the expected row ----
    t0 = time.perf_counter()
    if os.access(REF, os.X_OK):
        r = subprocess.run([REF, "cycle.fq.gz"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        ref_row, kind = r.stdout.decode().strip().splitlines()[-1].split("\t"), "reference binary (oracle/_ref/fastq_count)"
    else:
        ref_row, kind = None, "this build (oracle/_ref/fastq_count is absent: no independent row)"
    out["cycle_row"] = {"by": kind, "row": ref_row, "seconds": round(time.perf_counter() - t0, 2)}
    ours_cycle = subprocess.run([os.path.join(BIN, "fastq_count"), "cycle.fq.gz"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout.decode().strip().splitlines()[-1].split("\t")
    if ref_row is None:
        ref_row = ours_cycle
        out["cycle_row"]["row"] = ref_row
    want = ["c2.fq.gz", str(int(ref_row[1]) * cycles), str(int(ref_row[2]) * cycles)] + ref_row[3:]
    out["cycle_row_identical_to_ours"] = ours_cycle == ref_row
    # ---- the 1e9-read file ----
    runs = []
    for rep in range(3):
        time.sleep(20 if rep else 2)      # (the file was just written / the run before has just freed ~100 GB of device memory: let both settle)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(BIN, "fastq_count"), "c2.fq.gz"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": os.environ.get("C2_TIMING", "1")})
        dt = time.perf_counter() - t0
        row = p.stdout.decode().strip().splitlines()[-1].split("\t") if p.stdout.strip() else []
        runs.append({"seconds": round(dt, 2), "rc": p.returncode, "gbases_per_s": round(cycles * K * per * L / dt / 1e9, 2), "compressed_GBps": round(len(cycle) * cycles / dt / 1e9, 2),
                     "text_GBps": round(text_bytes * cycles / dt / 1e9, 2), "row": row, "row_identical": row == want,
                     "stderr": [l for l in p.stderr.decode().splitlines() if l.startswith("[hpn")][-int(os.environ.get("C2_LINES", "4")):]})
    out["expected_row"] = want
    out["runs"] = runs
    out["outputs_identical"] = all(r["row_identical"] and r["rc"] == 0 for r in runs) and out["cycle_row_identical_to_ours"]
    print(json.dumps(out))
    import shutil
    shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
