#!/usr/bin/env python3
"""The text front end on text that lies on the device (hpn_fastq_text_count_inplace: what the gzip route calls per 1 GiB of a
batch), for HIP-event timing and rocprofv3 passes:

    python3 scripts/bench_text_inplace.py [reads=3380000] [reps=5] [L=150]   -> one JSON line

`reads` x L bp as FASTQ text with 10-digit names (1 GiB at the defaults).  Counts are checked against the closed form of the
generator's total on every repetition (the same check bench.py makes); ms = whole call (lines + records + gather + tally) by host clock
around a synchronised call, and `kernel_ms` from the context's text family events where the build has them."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import highperformancengs_amd as hp  # noqa: E402


def device_text(ctx, n, L, seed=7, name_w=10):
    dq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_fastq_dev(seed, 0, n, L, dq, db, do)
    ctx.sync()
    rec = 2 + name_w + 1 + L + 1 + 2 + L + 1
    buf = torch.full((8192 + n * rec + 64,), 0x41, dtype=torch.uint8, device="cuda")
    t = buf[8192:8192 + n * rec].view(n, rec)
    t[:, 0], t[:, 1] = ord("@"), ord("r")
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    for k in range(name_w):
        t[:, 2 + k] = (48 + (idx // 10 ** (name_w - 1 - k)) % 10).to(torch.uint8)
    p = 2 + name_w
    t[:, p] = 10
    t[:, p + 1:p + 1 + L] = db.view(n, L)
    t[:, p + 1 + L] = 10
    t[:, p + 2 + L], t[:, p + 3 + L] = ord("+"), 10
    t[:, p + 4 + L:p + 4 + 2 * L] = dq.view(n, L)
    t[:, p + 4 + 2 * L] = 10
    qsum = int(dq.to(torch.int64).sum().item())
    del dq, db, do, idx
    torch.cuda.synchronize()
    return buf, n * rec, qsum


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_380_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    ctx = hp.Context(0)
    buf, nbytes, qsum = device_text(ctx, n, L)
    times = []
    for r in range(reps + 1):
        ctx.text_begin()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = ctx.text_count_inplace(buf[8192:], nbytes, last=True)
        ctx.sync()
        dt = (time.perf_counter() - t0) * 1e3
        assert not info.irregular and info.n_records == n, (info.irregular, info.n_records)
        res = ctx.fastq_tally_fetch()
        assert res.total == n * L, res.total
        if r:
            times.append(dt)
    times.sort()
    ms = times[len(times) // 2]
    print(json.dumps({"leg": "text framed in place", "reads": n, "L": L, "text_bytes": nbytes, "ms_call": round(ms, 4),
                      "GBps_call": round(nbytes / ms / 1e6, 1), "lib": os.environ.get("HPN_LIB", "tree")}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
