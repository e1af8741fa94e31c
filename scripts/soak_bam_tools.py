#!/usr/bin/env python3
"""Soak of the BAM tools (bam2depth -W, bam2wig, bam_sliding_count: BGZF inflate on the device, records read in place, depth /
window kernels, device-side bedGraph text) on random BAM FILES the tests do not hold: 1 - 4 targets with names of 1 - 60
characters, 2 - 40 K records with CIGARs put together from every operation (M I D N S H P = X) and lengths 1 - 300, all the
flag bits the filters look at, read names of 1 - 200 characters, every nibble code in the sequences, reads without a sequence
-- written samtools' way (records never cross a block) or htsjdk's (bamio.repack_bam, blocks of a random size), at a random
zlib level, run with random chunk / launch sizes and 1 - 3 workers.  Every output file must equal the oracle's text for the
records as the Python decoder (highperformancengs_amd/bamio.py) reads them back from the file.

    python3 scripts/soak_bam_tools.py [N=60] [first=0]  -> one JSON line (files first .. first + N - 1 of the seeded sequence)
    python3 scripts/soak_bam_tools.py N first damaged   -> the same files with the CONTAINER damaged (a flipped bit, a cut, bytes inserted):
        no reference behaviour to compare with (samtools' reader checks no CRC and walks whatever comes out); asked here: no tool dies of a
        signal or hangs, and where the device ingest and the host ingest (HPN_BAM_GPU=0) both finish with code 0 their outputs are the same"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from highperformancengs_amd import bamio  # noqa: E402

BIN = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")
CODES = "=ACMGRSVTWYHKDBN"


def random_cigar(rng):
    ops = "MIDNSHP=X"
    n = int(rng.integers(1, 7))
    out = []
    for k in range(n):
        op = "M" if rng.random() < 0.5 else ops[int(rng.integers(0, len(ops)))]
        ln = int(rng.integers(1, 12)) if op in "IDP" and rng.random() < 0.7 else int(rng.integers(1, 301))
        if op == "N":
            ln = int(rng.integers(1, 1500))
        out.append((ln, op))
    if not any(o in "MIS=X" for _, o in out):
        out.append((int(rng.integers(1, 150)), "M"))
    return out


def make_bam(rng, path):
    n_ref = int(rng.integers(1, 5))
    refs = []
    for t in range(n_ref):
        name = "".join(chr(int(c)) for c in rng.integers(48, 123, int(rng.integers(1, 61)))).replace("\\", "_").replace("`", "_")
        name = "".join(ch if ch.isalnum() or ch in "._-" else "_" for ch in name) + str(t)
        refs.append((name, int(rng.integers(3000, 2_500_000))))
    n = int(rng.integers(2000, 40000))
    lens = np.array([l for _, l in refs], np.float64)
    tid = np.sort(rng.choice(n_ref, size=n, p=lens / lens.sum()))
    cigs = [random_cigar(rng) for _ in range(int(rng.integers(3, 14)))]
    recs = []
    pos_all = np.zeros(n, np.int64)
    for t in range(n_ref):
        m = tid == t
        pos_all[m] = np.sort(rng.integers(0, refs[t][1], int(m.sum())))
    flag_pool = np.array([0, 16, 99, 147, 4, 256, 512, 1024, 2048, 1 | 64, 4 | 16, 256 | 16, 1024 | 99])
    for i in range(n):
        cg = cigs[int(rng.integers(0, len(cigs)))]
        qlen = sum(l for l, o in cg if o in "MIS=X")
        words = bamio.parse_cigar("".join(f"{l}{o}" for l, o in cg))
        no_seq = rng.random() < 0.02
        seq = "*" if no_seq else "".join(CODES[int(c)] for c in rng.integers(0, 16, qlen))
        qual = b"" if no_seq or rng.random() < 0.3 else bytes(rng.integers(0, 42, qlen, dtype=np.uint8))
        name = "r%d" % i + "x" * (int(rng.integers(0, 190)) if rng.random() < 0.05 else int(rng.integers(0, 20)))
        t = int(tid[i])
        if rng.random() < 0.01:                    # an unplaced read at the end of its target's records would break the order: keep tid, drop pos
            flag = 4
        else:
            flag = int(flag_pool[int(rng.integers(0, len(flag_pool)))])
        recs.append(bamio.BamRecord(tid=t, pos=int(pos_all[i]), flag=flag, cigar=words, seq=seq, qual=qual, name=name,
                                    mapq=int(rng.integers(0, 61)), mtid=t if rng.random() < 0.5 else -1,
                                    mpos=int(rng.integers(0, refs[t][1])) if rng.random() < 0.5 else -1, isize=int(rng.integers(-500, 500))))
    level = int(rng.integers(1, 10))
    bamio.write_bam(path, refs, recs, level=level)
    layout = "samtools"
    if rng.random() < 0.5:
        block = int(rng.choice([700, 3000, 20000, 65280]))
        bamio.repack_bam(path, path + ".packed", block, level=level)
        os.replace(path + ".packed", path), os.replace(path + ".packed.bai", path + ".bai")
        layout = f"packed {block}"
    return n, layout


def damaged_main(N, first):
    td = tempfile.mkdtemp(prefix="soak_bamd_")
    codes, same, differ_ok = {}, 0, 0
    for i in range(first, first + N):
        rng = np.random.default_rng(31_000 + i)
        d = os.path.join(td, "w")
        os.makedirs(d)
        bam = os.path.join(d, "s.bam")
        make_bam(rng, bam)
        b = bytearray(open(bam, "rb").read())
        k = int(rng.integers(0, 3))
        lo = len(b) // 10                                        # (behind the header's blocks)
        if k == 0:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(lo * 8, len(b) * 8))
                b[at >> 3] ^= 1 << (at & 7)
        elif k == 1:
            del b[int(rng.integers(lo, len(b))):]
        else:
            at = int(rng.integers(lo, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        open(bam, "wb").write(bytes(b))
        W = int(rng.choice([100, 1000, 20000]))
        outs = []
        for env in ({}, {"HPN_BAM_GPU": "0"}):
            e = {**os.environ, **env, "HPN_NGPU": str(int(rng.integers(1, 3)))}
            res = []
            for tool, args, files in (("bam2depth", ["-w", str(W), "-o", "d", "s.bam"], ["s.bam.1.bedGraph", "d.1.depth"]),
                                      ("bam_sliding_count", ["-w", str(W), "-o", "s", "s.bam"], ["s.txt"])):
                for f in files:
                    if os.path.exists(os.path.join(d, f)):
                        os.remove(os.path.join(d, f))
                p = subprocess.run([os.path.join(BIN, tool)] + args, cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
                assert p.returncode >= 0, (i, k, tool, env, "died of signal", -p.returncode, p.stderr.decode()[-800:])
                codes[p.returncode] = codes.get(p.returncode, 0) + 1
                res.append((p.returncode, [open(os.path.join(d, f), "rb").read() if os.path.exists(os.path.join(d, f)) else None for f in files]))
            outs.append(res)
        for a, b2 in zip(outs[0], outs[1]):
            if a[0] == 0 and b2[0] == 0:
                assert a[1] == b2[1], (i, k, "device and host ingest both finished, with different outputs")
                same += 1
            else:
                differ_ok += 1
        shutil.rmtree(d)
    os.rmdir(td)
    print(json.dumps({"damaged_bam_files": N, "first": first, "exit_codes": codes, "tool_runs_where_both_ingests_finished_and_agree": same,
                      "tool_runs_where_one_or_both_refused": differ_ok, "died_of_a_signal_or_hung": 0}))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "damaged":
        return damaged_main(N, first)
    td = tempfile.mkdtemp(prefix="soak_bam_")
    on_device = on_host = refused = regions = pairs = 0
    for i in range(first, first + N):
        rng = np.random.default_rng(31_000 + i)
        d = os.path.join(td, "w")
        os.makedirs(d)
        bam = os.path.join(d, "s.bam")
        n, layout = make_bam(rng, bam)
        soa = bamio.read_bam_records(bam)
        W = int(rng.choice([7, 37, 100, 1000, 20000, 65536]))
        env = {**os.environ, "HPN_TIMING": "1", "HPN_NGPU": str(int(rng.integers(1, 4)))}
        if rng.random() < 0.6:
            env["HPN_BAM_CHUNK"] = str(int(rng.integers(60_000, 2_000_000)))
            env["HPN_BAM_ROUNDS"] = str(int(rng.integers(1, 8)))
        knobs = {k: v for k, v in env.items() if k.startswith("HPN_BAM") or k == "HPN_NGPU"}
        what = (i, n, layout, W, knobs)
        try:
            bed, dep, wig, chrom = orc.bam2depth_text(soa, W, wig=True)
            domain = True
        except AssertionError:
            domain = False                       # (an M block beyond the reference's key range: the tool must say so, not compute)
        p = subprocess.run([os.path.join(BIN, "bam2depth"), "-w", str(W), "-W", "-o", "d", "s.bam"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=600)
        if not domain:
            assert p.returncode != 0, what
            refused += 1
        else:
            assert p.returncode == 0, (what, p.stderr.decode()[-1500:])
            for f, want in (("s.bam.1.bedGraph", bed), ("d.1.depth", dep), ("d.1.wig", wig), ("d.1.chromSize.txt", chrom)):
                assert open(os.path.join(d, f), "rb").read() == want, (what, f, p.stderr.decode()[-1500:])
            wwig, wchrom = orc.bam2wig_text(soa, W)
            p2 = subprocess.run([os.path.join(BIN, "bam2wig"), "-w", str(W), "-o", "w", "s.bam"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert p2.returncode == 0, (what, p2.stderr.decode()[-1500:])
            assert open(os.path.join(d, "w.1.wig"), "rb").read() == wwig and open(os.path.join(d, "w.1.chromSize.txt"), "rb").read() == wchrom, (what, "bam2wig")
        if True:                                 # (also where len / W reaches 65536: the window index wraps like the reference's unsigned short)
            want = orc.window_report(soa, W)
            p3 = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", str(W), "-o", "s", "s.bam"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                timeout=600)
            assert p3.returncode == 0, (what, p3.stderr.decode()[-1500:])
            assert open(os.path.join(d, "s.txt"), "rb").read() == want, (what, "bam_sliding_count", p3.stderr.decode()[-1500:])
        if rng.random() < 0.6:                   # -r: the records bam_fetch would hand over (index start, is_overlap), window report of those
            t = int(rng.integers(0, len(soa.refs)))
            name, tlen = soa.refs[t]
            a = int(rng.integers(0, tlen))
            b_ = int(rng.integers(a + 1, tlen + 1))
            want = orc.window_report(orc.region_subset(soa, t, a, b_), W)
            p4 = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", str(W), "-r", "%s:%d-%d" % (name, a + 1, b_), "-o", "reg", "s.bam"], cwd=d, env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert p4.returncode == 0, (what, "region", p4.stderr.decode()[-1500:])
            assert p4.stdout == b"%s\t%d\t%d\n" % (name.encode(), a, b_), (what, "region stdout", p4.stdout)
            assert open(os.path.join(d, "reg.txt"), "rb").read() == want, (what, "bam_sliding_count -r", name, a, b_, p4.stderr.decode()[-1500:])
            regions += 1
        if domain and rng.random() < 0.3:         # two inputs: bam2depth writes .1. and .2. files, bam_sliding_count reports the first input only
            make_bam(rng, os.path.join(d, "b.bam"))
            soa2 = bamio.read_bam_records(os.path.join(d, "b.bam"))
            try:
                bed2, dep2, _, _ = orc.bam2depth_text(soa2, W, wig=True)
            except AssertionError:
                bed2 = None
            if bed2 is not None:
                p5 = subprocess.run([os.path.join(BIN, "bam2depth"), "-w", str(W), "-o", "two", "s.bam", "b.bam"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                    timeout=600)
                assert p5.returncode == 0, (what, "two inputs", p5.stderr.decode()[-1500:])
                for f, want2 in (("s.bam.1.bedGraph", bed), ("two.1.depth", dep), ("b.bam.2.bedGraph", bed2), ("two.2.depth", dep2)):
                    assert open(os.path.join(d, f), "rb").read() == want2, (what, "two inputs", f)
                p6 = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-w", str(W), "-o", "two", "s.bam", "b.bam"], cwd=d, env=env, stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, timeout=600)
                assert p6.returncode == 0 and open(os.path.join(d, "two.txt"), "rb").read() == orc.window_report(soa, W), (what, "sliding, two inputs")
                pairs += 1
        if b"GPU ingest" in p.stderr and b"host ingest" not in p.stderr:
            on_device += 1
        else:
            on_host += 1
        shutil.rmtree(d)
    os.rmdir(td)
    print(json.dumps({"bam_files": N, "first": first, "ingested_on_the_device": on_device, "ingested_on_the_host": on_host, "outside_the_domain_and_refused": refused, "region_runs_(-r)_equal": regions, "runs_on_two_inputs_equal": pairs,
                      "outputs": "bedGraph, depth, wig, chromSize (bam2depth -W), wig + chromSize (bam2wig), out.txt (bam_sliding_count): all equal to the oracle's"}))


if __name__ == "__main__":
    main()
