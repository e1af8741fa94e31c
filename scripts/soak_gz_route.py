#!/usr/bin/env python3
"""Soak of the gzip route through the TOOL (fastq_count with the device inflate forced) on streams zlib can make and the
tests do not: every level 0..9, every strategy (default, filtered, Huffman only, RLE, fixed codes), memLevel 1..9, sync /
full flushes at random places, 1..4 members cut at arbitrary bytes -- under random stretch sizes, batch counts, searches on
the host / the device and framing slices.  The report must be the oracle's (the 4 x gzgets restatement over zlib's gzread),
byte for byte, whether the route takes the file or hands it back (fixed codes and stored blocks give the search nothing
to find: counted, not an error).

    python3 scripts/soak_gz_route.py [N=200] [first=0]  -> one JSON line"""
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")
STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]


def fastq(rng):
    n = int(rng.integers(4000, 60000))
    lo = int(rng.integers(30, 200))
    hi = lo if rng.random() < 0.6 else lo + int(rng.integers(1, 60))
    lens = rng.integers(lo, hi + 1, n)
    tot = int(lens.sum())
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), tot, p=[0.2475] * 4 + [0.01]).tobytes()
    qual = rng.integers(33, 75, tot, dtype=np.uint8).tobytes()
    out, at = [], 0
    tag = b"@" + bytes(rng.integers(65, 91, int(rng.integers(1, 30)), dtype=np.uint8)) + b":"
    for i in range(n):
        l = int(lens[i])
        out.append(b"%s%d\n%s\n+\n%s\n" % (tag, i, seq[at:at + l], qual[at:at + l]))
        at += l
    return b"".join(out)


def gz(rng, data):
    level, strategy, mem = int(rng.integers(0, 10)), STRATEGIES[int(rng.integers(0, 5))], int(rng.integers(1, 10))
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    out, at = [], 0
    flush_every = 0 if rng.random() < 0.5 else int(rng.integers(20_000, 2_000_000))
    while at < len(data):
        step = len(data) - at if not flush_every else min(len(data) - at, int(rng.integers(flush_every // 2 + 1, flush_every + 2)))
        out.append(c.compress(data[at:at + step]))
        at += step
        if flush_every and at < len(data):
            out.append(c.flush(zlib.Z_SYNC_FLUSH if rng.random() < 0.7 else zlib.Z_FULL_FLUSH))
    out.append(c.flush())
    return b"".join(out), (level, strategy, mem, flush_every)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    td = tempfile.mkdtemp(prefix="soak_gz_")
    taken = handed_back = 0
    why = {}
    for i in range(first, first + N):
        rng = np.random.default_rng(7_100_000 + i)
        text = fastq(rng)
        cuts = sorted(int(x) for x in rng.integers(1, len(text), int(rng.integers(0, 4))))
        parts, how = [], []
        for a, b in zip([0] + cuts, cuts + [len(text)]):
            blob, h = gz(rng, text[a:b])
            parts.append(blob), how.append(h)
        path = os.path.join(td, "s.fq.gz")
        open(path, "wb").write(b"".join(parts))
        want = orc.fastq_count_report([path], names=["s.fq.gz"], header=True, length_detail=True)
        env = {**os.environ, "HPN_TIMING": "1", "HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": str(int(rng.integers(30_000, 700_000))),
               "HPN_GZ_FIND": "host" if rng.random() < 0.4 else "device"}
        if rng.random() < 0.6:
            env["HPN_GZ_BATCH"] = str(int(rng.integers(2, 12)))
        if rng.random() < 0.5:
            env["HPN_TEXT_SLICE"] = str(int(rng.integers(4000, 3_000_000)))
        p = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", "s.fq.gz"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                           timeout=300)
        knobs = {k: v for k, v in env.items() if k.startswith("HPN_GZ") or k == "HPN_TEXT_SLICE"}
        assert p.returncode == 0, (i, how, knobs, p.stderr.decode()[-2000:])
        assert p.stdout == want, (i, how, knobs, p.stderr.decode()[-2000:])
        if b"[hpn] gzip on the GPU" in p.stderr:
            taken += 1
        else:
            handed_back += 1
            k = "a member of stored blocks or fixed codes" if any(h[0] == 0 or h[1] == zlib.Z_FIXED for h in how) else "other"
            why[k] = why.get(k, 0) + 1
    os.remove(path)
    os.rmdir(td)
    print(json.dumps({"files": N, "first": first, "inflated_on_the_device": taken, "handed_back_to_the_host_readers": handed_back,
                      "handed_back_because": why, "reports": "all equal to the oracle's"}))


if __name__ == "__main__":
    main()
