#!/usr/bin/env python3
"""Soak of the FASTQ tools on random inputs in every container: fastq_count -H -L, fastq_count_kthread -H -L (merged report and
the per-file .tsv with the Quality matrix) and fastq_trim -s S -e E on 1 - 4 files each -- plain text, gzip (any level /
strategy / flush pattern, 1 - 3 members) or BGZF (bgzip's blocks, a random payload size) --, regular FASTQ of fixed or ragged
read lengths, with LF or CRLF, with or without the final newline, or with one irregularity (a NUL, a lost / extra newline, a long
line, a cut-off record, a short quality line: tests/test_fastq_text_gpu.py::_mutate), under 1 - 4 lanes and random chunk sizes.
Every report, .tsv and trimmed file must equal the oracle's (the reference's loops restated over zlib's gzread).

    python3 scripts/soak_fastq_tools.py [N=100] [first=0]  -> one JSON line
    SOAK_TRIM_ANY_S=1: fastq_trim's -s drawn from 0 .. 119 whatever the reads' lengths (the stale-buffer quirk of readNextNode, fastq_trim.c:67-108)"""
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import orc  # noqa: E402
import soak_gz_route as G  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")


def bgzf(rng, data):
    size = int(rng.choice([500, 4000, 30000, 65280]))
    out = []
    for a in range(0, len(data), size):
        piece = data[a:a + size]
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        comp = co.compress(piece) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                   struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))     # the empty block at the end
    return b"".join(out)


def mutate(rng, text):
    kind = int(rng.integers(0, 7))
    b = bytearray(text)
    nls = np.flatnonzero(np.frombuffer(text, np.uint8) == 10)
    if kind == 0:
        del b[int(rng.choice(nls))]
    elif kind == 1:
        b[int(rng.integers(0, len(b)))] = 0
    elif kind == 2:
        i = int(rng.integers(0, len(b)))
        b[i:i] = b"A" * int(rng.integers(1023, 3000))
    elif kind == 3:
        del b[int(rng.integers(1, len(b))):]
    elif kind == 4:
        b.insert(int(rng.integers(0, len(b))), 10)
    elif kind == 5 and len(nls) > 8:
        k = int(rng.integers(0, len(nls) // 4)) * 4 + 3
        if nls[k] - nls[k - 1] > 3:
            del b[int(nls[k]) - 2:int(nls[k])]
    else:
        b += b"@partial"
    return bytes(b)


def one_file(rng, path_stem):
    text = G.fastq(rng)
    if len(text) > 6_000_000:
        text = text[:text.rfind(b"\n@", 0, 6_000_000) + 1]
    what = ["regular"]
    r = rng.random()
    if r < 0.10:
        text = text.replace(b"\n", b"\r\n")
        what = ["crlf"]
    elif r < 0.20:
        text = text[:-1]
        what = ["no final newline"]
    elif r < 0.45:
        text = mutate(rng, text)
        what = ["damaged"]
    c = rng.random()
    if c < 0.4:
        path, blob = path_stem + ".fq", text
    elif c < 0.75:
        cuts = sorted(int(x) for x in rng.integers(1, max(2, len(text)), int(rng.integers(0, 3))))
        blob = b"".join(G.gz(rng, text[a:b])[0] for a, b in zip([0] + cuts, cuts + [len(text)]))
        path = path_stem + ".fq.gz"
        what.append("gzip")
    else:
        path, blob = path_stem + ".bgz.fq.gz", bgzf(rng, text)
        what.append("bgzf")
    if c >= 0.4 and rng.random() < 0.2:          # the CONTAINER damaged: the tools must deliver what zlib's gzread delivers of it
        b = bytearray(blob)
        k = int(rng.integers(0, 4))
        if k == 0:
            at = int(rng.integers(0, len(b) * 8))
            b[at >> 3] ^= 1 << (at & 7)
            what.append("a bit flipped in the container")
        elif k == 1:
            del b[int(rng.integers(1, len(b))):]
            what.append("container cut short")
        elif k == 2:
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8))
            what.append("bytes behind the container")
        else:
            at = int(rng.integers(0, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 50)), dtype=np.uint8))
            what.append("bytes inserted into the container")
        blob = bytes(b)
    open(path, "wb").write(blob)
    return path, what


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    td = tempfile.mkdtemp(prefix="soak_fq_")
    runs = {"fastq_count": 0, "fastq_count_kthread": 0, "fastq_trim": 0}
    kinds = {}
    for i in range(first, first + N):
        rng = np.random.default_rng(55_000 + i)
        d = os.path.join(td, "w")
        os.makedirs(d)
        nf = int(rng.integers(1, 5))
        files, whats = [], []
        for k in range(nf):
            p, w = one_file(rng, os.path.join(d, "f%d" % k))
            files.append(os.path.basename(p)), whats.append(w)
            kinds[" ".join(w)] = kinds.get(" ".join(w), 0) + 1
        env = {**os.environ, "HPN_TIMING": "1"}
        if rng.random() < 0.5:
            env["HPN_NGPU"] = str(int(rng.integers(1, 5)))
        if rng.random() < 0.5:
            env["HPN_TEXT_CHUNK"] = str(int(rng.integers(1 << 16, 1 << 23)))
        if rng.random() < 0.3:
            env["HPN_GZ_GPU_FORCE"] = "1"
            env["HPN_GZ_STRETCH"] = str(int(rng.integers(30_000, 400_000)))
        knobs = {k: v for k, v in env.items() if k.startswith("HPN_") and k != "HPN_TIMING"}
        what = (i, files, whats, knobs)
        paths = [os.path.join(d, f) for f in files]
        # a read of 512 bases or more (a damaged text can make one) is outside fastq_count's SeqLen[512]: the oracle says so (-2) and
        # the tools leave with exit code 2 and no report
        long_read = any(orc.count_stream(q)[0] != 0 for q in paths)
        # ---- fastq_count: one thread, rows in input order ----
        p = subprocess.run([os.path.join(BIN, "fastq_count"), "-t", "1", "-H", "-L"] + files, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        pk = subprocess.run([os.path.join(BIN, "fastq_count_kthread"), "-t", str(int(rng.integers(1, 5))), "-H", "-L", "-o", "m.tsv"] + files, cwd=d, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        if long_read:
            assert p.returncode == 2 and pk.returncode == 2, (what, p.returncode, pk.returncode)
            runs["refused: a read of 512 bases or more"] = runs.get("refused: a read of 512 bases or more", 0) + 1
        else:
            want = orc.fastq_count_report(paths, names=files, header=True, length_detail=True)
            assert p.returncode == 0, (what, p.stderr.decode()[-1500:])
            assert p.stdout == want, (what, "fastq_count", p.stderr.decode()[-1500:])
            runs["fastq_count"] += 1
            # ---- fastq_count_kthread: merged report + per-file tsv ----
            merged, per_file = orc.kthread_report(paths, names=files, header=True, length_detail=True)
            assert pk.returncode == 0, (what, pk.stderr.decode()[-1500:])
            assert open(os.path.join(d, "m.tsv"), "rb").read() == merged, (what, "kthread merged", pk.stderr.decode()[-1500:])
            for k, f in enumerate(files):
                assert open(os.path.join(d, "%s.%d.tsv" % (f, k)), "rb").read() == per_file[k], (what, "kthread tsv", f)
            runs["fastq_count_kthread"] += 1
        # ---- fastq_trim on the first file (S inside every read: beyond a read's end the reference copies stale bytes) ----
        lo = min((len(l) for l in open(paths[0], "rb").read().split(b"\n")[1::4]), default=0) if files[0].endswith(".fq") and whats[0] == ["regular"] else 0
        S = int(rng.integers(0, min(lo, 40) + 1)) if lo else 0
        if os.environ.get("SOAK_TRIM_ANY_S") and "damaged" not in whats[0]:     # also S beyond a read's end: the reference copies stale buffer bytes there, and so must the tool
            S = int(rng.integers(0, 120))
        E = int(rng.integers(S + 1, 320))
        rc, wtext, nw = orc.trim_stream(paths[0], S, E)
        p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", files[0], "-o", "t", "-s", str(S), "-e", str(E)], cwd=d, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        if rc == 0:
            assert p.returncode == 0, (what, S, E, p.stderr.decode()[-1500:])
            assert open(os.path.join(d, "t.trim.fastq"), "rb").read() == wtext, (what, "fastq_trim", S, E, p.stderr.decode()[-1500:])
            assert b"Total_reads: %d\n" % nw in p.stderr, (what, p.stderr[-300:])
            runs["fastq_trim"] += 1
        # ---- the pipes: fastq_trim from stdin to stdout, fastq_count from stdin (a sound container only: output that cannot be rewound
        # is refused for a damaged one, tests/test_cli_gpu.py::test_trim_of_a_damaged_gzip) ----
        if rc == 0 and not any("container" in w for w in whats[0]) and rng.random() < 0.5:
            with open(paths[0], "rb") as fh:
                p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-s", str(S), "-e", str(E)], cwd=d, env=env, stdin=fh, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   timeout=600)
            assert p.returncode == 0 and p.stdout == wtext, (what, "fastq_trim stdin -> stdout", S, E, p.returncode, p.stderr.decode()[-800:])
            runs["fastq_trim through pipes"] = runs.get("fastq_trim through pipes", 0) + 1
            if not long_read:
                with open(paths[0], "rb") as fh:
                    p = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", "-"], cwd=d, env=env, stdin=fh, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                assert p.returncode == 0 and p.stdout == orc.fastq_count_report([paths[0]], names=["-"], header=True, length_detail=True), (what, "fastq_count -")
                runs["fastq_count from stdin"] = runs.get("fastq_count from stdin", 0) + 1
        shutil.rmtree(d)
    os.rmdir(td)
    print(json.dumps({"rounds": N, "first": first, "tool_runs_compared": runs, "files_by_kind": kinds, "outputs": "all equal to the oracle's"}))


if __name__ == "__main__":
    main()
