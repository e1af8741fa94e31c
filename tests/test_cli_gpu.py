"""GPU: the CLI tools are drop-ins -- same argv, same stdout, same files, byte for byte,
as the compiled reference tools recorded in tests/golden/ (make_golden.py)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import orc
from conftest import GOLDEN, expected, golden_path
from highperformancengs_amd import bamio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "highperformancengs_amd", "bin")

CASES = ["count_a1", "count_a1_gz", "count_empty", "count_nonl", "count_crlf", "count_multi", "count_short",
         "count_len0", "count_allzero", "count_trunc", "count_longname", "count_syn_var_a", "count_syn_var_b",
         "count_syn_100", "count_to_file", "kthread_a1", "kthread_syn", "kthread_plain", "kthread_empty",
         "trim_a1", "trim_a1_default", "trim_a1_file", "trim_nonl", "trim_short", "trim_crlf", "trim_syn_var",
         "trim_syn_100", "trim_multi", "trim_empty", "trim_stale_8_40", "trim_stale_15_400", "trim_stale_30_31",
         "trim_stale_a1", "count_stale",
         # damaged gzip members (CRC-32 / ISIZE): the reference counts what zlib's gzgets hands out before it fails
         "count_badcrc", "count_badcrc_mid", "count_badisize", "kthread_badcrc",
         "depth_a3", "depth_a3_wig", "depth_a3_stdout", "depth_rand", "depth_rand_w1000", "depth_two_files",
         "wig_a3", "wig_a3_w7", "wig_rand", "wig_rand_w1000", "wig_rand_w37",
         "sliding_a3", "sliding_rand", "sliding_rand_w700", "sliding_rand_w37", "sliding_region", "sliding_region_chr",
         "sliding_two_files", "sliding_two_files_rev", "sliding_wrap",
         # the command-line surface: stdin ("-", IO_stream.h:122-136), a missing input (O_CREAT makes it, :127), -h / unknown
         # option / no arguments (usage on stderr, exit 1: fastq_count.c:135-156,194-196), -v -z ignored (fastq_trim.c:133-138), five files on
         # three threads (rows in completion order: compared as a sorted set), bam2depth -r running into -s (bam2depth.c:281-285)
         "count_stdin", "count_stdin_gz", "count_stdin_mixed", "kthread_stdin", "trim_stdin", "trim_stdin_default_in",
         "count_missing", "kthread_missing", "trim_missing", "count_help", "count_badopt", "count_noargs", "kthread_help",
         "trim_help", "trim_noargs", "trim_badopt", "trim_ignored_v_z", "depth_help", "depth_noargs", "sliding_help", "wig_help",
         "count_t3_five", "kthread_t3_five", "depth_r_falls_into_s", "depth_r_only"]


# cases that end in the usage text before any route is chosen: run once (test_drop_in), not once per route
NO_ROUTE = ("_help", "_noargs", "_badopt")
FASTQ_ROUTE_CASES = [c for c in CASES if c.startswith(("count_", "kthread_", "trim_")) and not c.endswith(NO_ROUTE)]
BAM_ROUTE_CASES = [c for c in CASES if c.startswith(("depth_", "wig_", "sliding_")) and not c.endswith(NO_ROUTE)]


def _run(tool, args, inputs, cwd, env=None, stdin=None, bindir=BIN):
    for src in inputs:
        shutil.copy(src, cwd)
        if src.endswith(".bam"):
            shutil.copy(src + ".bai", cwd)
    before = set(os.listdir(cwd))
    p = subprocess.run([os.path.join(bindir, tool)] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, **(env or {})}, stdin=open(stdin, "rb") if stdin else subprocess.DEVNULL)
    return p, sorted(set(os.listdir(cwd)) - before)


def _check(manifest, case, tmp_path, env=None, force_t1=True, bindir=BIN):
    """Run golden case `case` through the tool here and compare return code, stdout and every file with the reference's."""
    c = manifest[case]
    args = list(c["args"])
    if force_t1 and c["tool"] == "fastq_count" and "-t" not in args and c["inputs"]:
        args = ["-t", "1"] + args  # rows are printed in completion order; one at a time = input order
    p, files = _run(c["tool"], args, [os.path.join(GOLDEN, i) for i in c["inputs"]], tmp_path, env,
                    os.path.join(GOLDEN, c["stdin"]) if c.get("stdin") else None, bindir)
    assert p.returncode == c["returncode"], p.stderr.decode()
    got = b"".join(sorted(p.stdout.splitlines(keepends=True))) if c.get("unordered") else p.stdout
    assert got == expected(case), p.stderr.decode()
    assert files == c["files"]
    for f in files:
        assert open(tmp_path / f, "rb").read() == expected(case, f), f
    if c.get("stderr_usage"):
        assert b"sage" in p.stderr           # the usage text went to stderr (its wording is this build's own)
    return p


@pytest.mark.parametrize("case", CASES)
def test_drop_in(manifest, case, tmp_path):
    _check(manifest, case, tmp_path)


# Half of the route coverage below runs on the test-hooks build (the same sources with -DHPN_TEST_HOOKS: csrc/host/knobs.hpp;
# tests/conftest.py swaps it in whenever a test sets one of the 31 switches).  With NO switch set that build must be the shipped
# one: every golden case through testhooks/bin gives the reference's bytes as well, so whatever a switched route does differently
# comes from test_env() alone.  (no reference counterpart.)
@pytest.mark.parametrize("case", CASES)
def test_hooks_build_with_no_switch_set_is_the_shipped_one(manifest, case, tmp_path):
    from conftest import HOOKS_BIN, TEST_KNOBS
    clean = {k: v for k, v in os.environ.items() if k not in TEST_KNOBS}
    c = manifest[case]
    args = list(c["args"])
    if c["tool"] == "fastq_count" and "-t" not in args and c["inputs"]:
        args = ["-t", "1"] + args
    for src in [os.path.join(GOLDEN, i) for i in c["inputs"]]:
        shutil.copy(src, tmp_path)
        if src.endswith(".bam"):
            shutil.copy(src + ".bai", tmp_path)
    before = set(os.listdir(tmp_path))
    stdin = open(os.path.join(GOLDEN, c["stdin"]), "rb") if c.get("stdin") else subprocess.DEVNULL
    p = subprocess.Popen([os.path.join(HOOKS_BIN, c["tool"])] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=stdin, env=clean)
    out, errb = p.communicate()
    files = sorted(set(os.listdir(tmp_path)) - before)
    assert p.returncode == c["returncode"], errb.decode()
    got = b"".join(sorted(out.splitlines(keepends=True))) if c.get("unordered") else out
    assert got == expected(case), errb.decode()
    assert files == c["files"]
    for f in files:
        assert open(tmp_path / f, "rb").read() == expected(case, f), f


# The FASTQ tools frame regular text on the GPU (hpn_fastq_text_*) and everything else with
# the exact gzgets emulation on the host: both routes, and a chunk size that cuts records
# every few bytes, must give the reference's bytes.
@pytest.mark.parametrize("env", [{"HPN_TEXT": "0"}, {"HPN_TEXT_CHUNK": "100"}, {"HPN_TEXT_CHUNK": "4099"},
                                 {"HPN_PGZ_FORCE": "1", "HPN_PGZ_CHUNK": "1500", "HPN_GZ_THREADS": "3"}],
                         ids=["host-framer", "chunk100", "chunk4099", "two-pass-gzip"])
@pytest.mark.parametrize("case", FASTQ_ROUTE_CASES)
def test_drop_in_fastq_routes(manifest, case, env, tmp_path):
    _check(manifest, case, tmp_path, env)


# ONE FASTQ input over several GPUs (host/text_shard.hpp): the input's bytes go in pieces cut anywhere to one
# context per device, every lane frames and tallies / trims the records its pieces own, the lanes' count vectors are summed
# where reduceStats sums files (fastq_count_kthread.c:180-210), trimmed slabs are written in piece order.  HPN_NGPU forces
# that many lanes on whatever devices exist, so the route runs on a one-GPU box: the bytes must be the reference's, and
# regular plain / host-inflated text must really have taken the route.
# (the damaged gzip files leave the route -- the reader reports the damage -- and are read again through zlib's own reader)
SHARDED_REGULAR = {"count_a1", "count_a1_gz", "count_empty", "count_crlf", "count_multi", "count_syn_var_a", "count_syn_var_b",
                   "count_syn_100", "count_to_file", "kthread_a1", "kthread_syn", "kthread_plain", "kthread_empty",
                   "trim_a1_file", "trim_syn_100"}


@pytest.mark.parametrize("env", [{"HPN_NGPU": "2"}, {"HPN_NGPU": "3", "HPN_TEXT_CHUNK": "8192"}, {"HPN_NGPU": "5", "HPN_TEXT_CHUNK": "20000"},
                                 {"HPN_NGPU": "8", "HPN_TEXT_CHUNK": "8192"}],     # the 8-way shape of a full node: 8 contexts, 10 pinned buffers, 8 tickets on the board
                         ids=["2lanes", "3lanes-8k", "5lanes-20k", "8lanes-8k"])
@pytest.mark.parametrize("case", FASTQ_ROUTE_CASES)
def test_drop_in_one_fastq_over_several_lanes(manifest, case, env, tmp_path):
    c = manifest[case]
    args = c["args"]
    p = _check(manifest, case, tmp_path, {**env, "HPN_TIMING": "1"})
    took = p.stderr.count(f"one input over {env['HPN_NGPU']} lanes".encode())
    if case in SHARDED_REGULAR:
        assert took >= 1 and b"abandoned" not in p.stderr, p.stderr.decode()
        assert b"(lanes share a device)" in p.stderr       # and said how the sum was made
    if c["tool"] == "fastq_trim" and "-o" not in args:
        assert took == 0                                    # output to stdout cannot be rewound: one context


# ONE gzip input over several GPUs (host/gz_shard.hpp): batches of deflate stretches go to the lanes in turn, every lane finds its
# own block starts, inflates symbolically, and three chains carry the 32 KiB window, the member state (ISIZE / CRC-32 as gzread
# checks them behind the reference's gzgets, IO_stream.h:122-136) and the line count from batch to batch.  Small stretches, batches
# and text slices put many seams into the goldens' small files; the damaged files must leave the route and still give the
# reference's bytes (what zlib hands out before it fails).
GZ_SHARDED = ["count_a1_gz", "count_multi", "count_syn_var_b", "count_syn_100", "kthread_a1", "kthread_syn", "count_badcrc", "count_badcrc_mid",
              "count_badisize", "kthread_badcrc", "count_stdin_gz", "count_t3_five"]


@pytest.mark.parametrize("env", [{"HPN_NGPU": "2", "HPN_GZ_STRETCH": "8192", "HPN_GZ_BATCH": "7", "HPN_TEXT_SLICE": "20000"},
                                 {"HPN_NGPU": "3", "HPN_GZ_STRETCH": "16384", "HPN_GZ_BATCH": "3", "HPN_GZ_FIND": "device"},
                                 {"HPN_NGPU": "8", "HPN_GZ_STRETCH": "16384", "HPN_GZ_BATCH": "4", "HPN_GZ_FIND": "host", "HPN_TEXT_SLICE": "9000"}],
                         ids=["2lanes", "3lanes-device-search", "8lanes"])
@pytest.mark.parametrize("case", GZ_SHARDED)
def test_drop_in_one_gzip_over_several_lanes(manifest, case, env, tmp_path):
    p = _check(manifest, case, tmp_path, {**env, "HPN_GZ_GPU_FORCE": "1", "HPN_TIMING": "1"})
    took = p.stderr.count(f"one gzip input over {env['HPN_NGPU']} lanes".encode())
    if "bad" in case:
        assert b"gzip over " in p.stderr and b"abandoned" in p.stderr, p.stderr.decode()      # the member checks saw the damage
    elif case in ("count_multi", "count_syn_var_b", "count_syn_100", "kthread_syn", "count_t3_five"):
        assert took >= 1, p.stderr.decode()


def test_sharded_route_on_a_larger_file(tmp_path):
    """5e5 reads (~160 MB): 4 lanes, 1 MiB pieces -> ~160 pieces racing through the board; the three tools' outputs equal
    the one-context route's, which the goldens pin to the reference."""
    n, L = 500000, 150
    seq, qual, off = orc.synth_soa(777, 0, n, L, L)
    s, q = seq.reshape(n, L), qual.reshape(n, L)
    with open(tmp_path / "big.fq", "wb") as fh:
        for i in range(n):
            fh.write(b"@read%d/1\n" % i + s[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
    outs = {}
    for tag, env in (("one", {}), ("four", {"HPN_NGPU": "4", "HPN_TEXT_CHUNK": str(1 << 20), "HPN_TIMING": "1"})):
        d = tmp_path / tag
        d.mkdir()
        e = {**os.environ, **env}
        a = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", "../big.fq"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        b = subprocess.run([os.path.join(BIN, "fastq_count_kthread"), "-L", "../big.fq"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        t = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", "../big.fq", "-s", "5", "-e", "140", "-o", "t"], cwd=d, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=e)
        assert a.returncode == 0 and b.returncode == 0 and t.returncode == 0, (a.stderr, b.stderr, t.stderr)
        if tag == "four":
            for r in (a, b, t):
                assert b"one input over 4 lanes" in r.stderr and b"abandoned" not in r.stderr, r.stderr.decode()
        outs[tag] = (a.stdout, b.stdout, open(d / "big.fq.0.tsv", "rb").read(), open(d / "t.trim.fastq", "rb").read(),
                     [l for l in t.stderr.split(b"\n") if l.startswith(b"Total_reads")])
    assert outs["one"] == outs["four"]
    assert outs["one"][4] == [b"Total_reads: %d" % n]
    row = outs["one"][0].decode().split("\n")[1].split("\t")
    assert int(row[1]) == n and int(row[2]) == n * L


def test_trim_of_a_damaged_gzip(tmp_path):
    """The reference's fastq_trim dies with SIGSEGV on a gzip file whose member fails its CRC-32 (it dereferences the NULL of the
    failed gzgets): no golden.  Here: output to a file is made again through zlib's own reader once the damage shows (what zlib
    hands out before it fails, trimmed); output to stdout cannot be taken back and is refused with exit code 2."""
    src = golden_path("fastq", "badcrc_mid.fq.gz")
    p, files = _run("fastq_trim", ["-i", "badcrc_mid.fq.gz", "-s", "0", "-e", "100"], [src], tmp_path)
    assert p.returncode == 2 and b"damaged gzip" in p.stderr, p.stderr.decode()
    p, files = _run("fastq_trim", ["-i", "badcrc_mid.fq.gz", "-s", "0", "-e", "100", "-o", "bc"], [src], tmp_path)
    assert p.returncode == 0 and files == ["bc.trim.fastq"], p.stderr.decode()
    a = open(tmp_path / "bc.trim.fastq", "rb").read()
    # the records zlib's gzgets hands out whole before the failing buffer: 448 of them (the reference's fastq_count sees 449 reads,
    # the last one from stale buffer bytes: golden count_badcrc_mid)
    assert a.count(b"\n+\n") >= 448 and a.startswith(b"@r0\n")
    os.unlink(tmp_path / "bc.trim.fastq")
    p, files = _run("fastq_trim", ["-i", "badcrc_mid.fq.gz", "-s", "0", "-e", "100", "-o", "bc"], [src], tmp_path, {"HPN_TEXT": "0"})
    assert p.returncode == 0 and open(tmp_path / "bc.trim.fastq", "rb").read() == a


# The BAM tools inflate BGZF blocks and walk the records on the GPU when every block starts at a
# record boundary (as samtools writes them), else on the host: both routes, and compressed
# chunks that cut blocks every 64 KiB, must give the reference's bytes.
@pytest.mark.parametrize("env", [{"HPN_BAM_GPU": "0"}, {"HPN_BAM_CHUNK": "65600"}, {"HPN_BEDGRAPH_HOST": "1"}],
                         ids=["host-ingest", "chunk64k", "bedgraph-from-runs"])
@pytest.mark.parametrize("case", BAM_ROUTE_CASES)
def test_drop_in_bam_routes(manifest, case, env, tmp_path):
    _check(manifest, case, tmp_path, env)


# Several GPUs: bam2depth / bam2wig hand whole targets, largest first, to one worker per device and write the
# results in target order; bam_sliding_count hands record batches to the devices in turn and adds their per-window
# vectors before the float32 replay.  HPN_NGPU forces that many workers on whatever devices exist (worker % devices), so
# the path runs on a one-GPU box: the bytes must be the reference's, and the route must really have been taken.
@pytest.mark.parametrize("env", [{"HPN_NGPU": "2"}, {"HPN_NGPU": "3", "HPN_BAM_CHUNK": "65600"}, {"HPN_NGPU": "8", "HPN_BAM_CHUNK": "65600"}],
                         ids=["2workers", "3workers-chunk64k", "8workers-chunk64k"])
@pytest.mark.parametrize("case", BAM_ROUTE_CASES)
def test_drop_in_bam_multi_gpu_route(manifest, case, env, tmp_path):
    c = manifest[case]
    p = _check(manifest, case, tmp_path, {**env, "HPN_TIMING": "1"})
    if "-r" not in c["args"] and c["inputs"]:      # (a region goes through the index on the host reader)
        # bam2depth / bam2wig give a worker to each target that holds records at most; bam_sliding_count uses all it is given
        assert b"GPU ingest on " in p.stderr and b" workers" in p.stderr and b"abandoned" not in p.stderr, p.stderr.decode()
        if env["HPN_NGPU"] != "8" or c["tool"] == "bam_sliding_count":
            assert f"GPU ingest on {env['HPN_NGPU']} workers".encode() in p.stderr, p.stderr.decode()


@pytest.mark.parametrize("case", ["depth_rand", "depth_two_files", "wig_rand"])
def test_depth_look_ahead_budget_holds_workers_back(manifest, case, tmp_path):  # noqa: D401
    """Finished targets wait in host memory until every earlier one is written; HPN_DEPTH_LOOKAHEAD bounds the estimated bytes of
    what is claimed and not yet written.  With a budget of one byte only the target the writer waits for (or any, when nothing is
    held) may be taken: the workers wait for the writer, and the bytes are still the reference's (round-3 advisor: the budget
    could not bound anything -- the lowest unclaimed target was always admitted).  bam2depth.c:325-339 is the loop this orders."""
    c = manifest[case]
    p, files = _run(c["tool"], list(c["args"]), [os.path.join(GOLDEN, i) for i in c["inputs"]], tmp_path,
                    {"HPN_NGPU": "3", "HPN_DEPTH_LOOKAHEAD": "1", "HPN_TIMING": "1"})
    assert p.returncode == c["returncode"], p.stderr.decode()
    assert p.stdout == expected(case) and files == c["files"]
    for f in files:
        assert open(tmp_path / f, "rb").read() == expected(case, f), f
    import re
    m = re.findall(rb"look-ahead: at most ([0-9.]+) MB of estimated output claimed and not yet written \(budget ([0-9.]+) MB\), (\d+) waits", p.stderr)
    assert m, p.stderr.decode()
    for peak, budget, waits in m:
        assert int(waits) > 0 and float(budget) < 1e-3       # workers were held back: one target at a time


def test_packed_bam_on_several_workers(tmp_path):
    """Records packed across BGZF blocks (htsjdk's way; bamio.repack_bam writes the matching .bai): bam2depth's workers each read
    their targets from the index offset -- which now points INTO a block -- and carry unfinished records from launch to launch;
    bam_sliding_count's batches-in-turn route cannot carry, gives the file back and the one-stream route decodes it on the GPU."""
    src = golden_path("bam", "rand.bam")
    for block in (20000, 777):
        d = tmp_path / str(block)
        d.mkdir()
        bamio.repack_bam(src, str(d / "rand.bam"), block, level=0)         # (level 0: ~410 KB, several 65 KB chunks)
        for tool, args, case, outs in (("bam2depth", ["-o", "r", "rand.bam"], "depth_rand", ["rand.bam.1.bedGraph", "r.1.depth"]),
                                       ("bam_sliding_count", ["rand.bam"], "sliding_rand", ["out.txt"])):
            for env in ({}, {"HPN_BAM_CHUNK": "65600", "HPN_BAM_ROUNDS": "1"}):
                p = subprocess.run([os.path.join(BIN, tool)] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env={**os.environ, "HPN_TIMING": "1", "HPN_NGPU": "2", **env})
                assert p.returncode == 0, p.stderr.decode()
                assert b"host ingest" not in p.stderr, p.stderr.decode()
                if tool == "bam2depth":
                    assert b"GPU ingest on 2 workers" in p.stderr and b"abandoned" not in p.stderr, p.stderr.decode()
                elif env:         # batches of one chunk in turn: records run from a batch into the next worker's
                    assert b"workers  (abandoned)" in p.stderr and b"[hpn] GPU ingest\n" in p.stderr, p.stderr.decode()
                else:             # the whole file is one batch: nothing to carry
                    assert b"[hpn] GPU ingest on 2 workers\n" in p.stderr, p.stderr.decode()
                for f in outs:
                    assert open(d / f, "rb").read() == expected(case, f), (tool, f, block, env)


def test_region_reads_from_the_index_offset(tmp_path):
    """bam_sliding_count -r: the records come from where the .bai's linear index puts the region's start (not from the top
    of the file) and reading stops behind the region's end; the report is the reference's (golden sliding_region*)."""
    src = golden_path("bam", "rand.bam")
    for args, case in ((["-w", "1000", "-r", "chr2:1,001-20000", "-o", "reg", "rand.bam"], "sliding_region"),
                       (["-w", "5000", "-r", "chr1", "-o", "reg", "rand.bam"], "sliding_region_chr"),
                       (["-w", "1000", "-r", "chr2:49,000-60,000", "-o", "reg", "rand.bam"], None),
                       (["-w", "1000", "-r", "chrE", "-o", "reg", "rand.bam"], None)):          # a target without records
        p, files = _run("bam_sliding_count", args, [src], tmp_path, {"HPN_TIMING": "1"})
        assert p.returncode == 0, p.stderr.decode()
        line = [l for l in p.stderr.split(b"\n") if l.startswith(b"[hpn] region: reading from virtual offset")]
        assert len(line) == 1
        off = int(line[0].split()[-1])
        if b"chr2" in " ".join(args).encode():
            assert off >> 16 > 0            # chr2's records lie blocks behind the header
        if case:
            assert p.stdout == expected(case) and open(tmp_path / "reg.txt", "rb").read() == expected(case, "reg.txt")
        else:   # no golden: the whole-file oracle restricted to the region
            soa = bamio.read_bam_records(src)
            ref, beg, end = (1, 48999, 60000) if "chr2:49,000-60,000" in args else (2, 0, 1 << 29)
            assert open(tmp_path / "reg.txt", "rb").read() == orc.window_report(orc.region_subset(soa, ref, beg, end), 1000)
        for f in files:
            os.unlink(tmp_path / f)
        os.unlink(tmp_path / "rand.bam"), os.unlink(tmp_path / "rand.bam.bai")


def test_bam_gpu_ingest_takes_both_writers_files_and_falls_back_on_damage(tmp_path):
    """HPN_TIMING names the ingest: golden BAMs (samtools: record-aligned blocks) decode on the GPU, and so do the same records
    packed across block boundaries (htsjdk) -- with launches small enough that records also run from one launch into the next;
    a file that ends inside a record is left to the host reader, which ends where bam_read1 does."""
    src = golden_path("bam", "rand.bam")
    p, _ = _run("bam2depth", ["-o", "d", "rand.bam"], [src], tmp_path, {"HPN_TIMING": "1"})
    assert p.returncode == 0 and b"[hpn] GPU ingest" in p.stderr and b"abandoned" not in p.stderr
    want = open(tmp_path / "rand.bam.1.bedGraph", "rb").read()
    want_s = expected("sliding_rand", "out.txt")
    for block, env in ((20000, {}), (20000, {"HPN_BAM_CHUNK": "65600", "HPN_BAM_ROUNDS": "1"}), (333, {"HPN_BAM_CHUNK": "65600", "HPN_BAM_ROUNDS": "1"}),
                       (65000, {"HPN_BAM_AHEAD": "0", "HPN_BAM_CHUNK": "65600", "HPN_BAM_ROUNDS": "1"}), (97, {})):
        d2 = tmp_path / ("packed%d%s" % (block, "".join(env.values())))
        d2.mkdir()
        bamio.repack_bam(src, str(d2 / "rand.bam"), block, level=0 if env else 6)      # (level 0: the file spans several chunks)
        p = subprocess.run([os.path.join(BIN, "bam2depth"), "-o", "d", "rand.bam"], cwd=d2, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env={**os.environ, "HPN_TIMING": "1", "HPN_NGPU": "1", **env})
        assert p.returncode == 0, p.stderr.decode()
        assert b"[hpn] GPU ingest" in p.stderr and b"abandoned" not in p.stderr and b"host ingest" not in p.stderr, p.stderr.decode()
        assert open(d2 / "rand.bam.1.bedGraph", "rb").read() == want, (block, env)
        p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "rand.bam"], cwd=d2, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env={**os.environ, "HPN_TIMING": "1", "HPN_NGPU": "1", **env})
        assert p.returncode == 0, p.stderr.decode()
        assert b"[hpn] GPU ingest" in p.stderr and b"abandoned" not in p.stderr, p.stderr.decode()
        assert open(d2 / "out.txt", "rb").read() == want_s, (block, env)
    # the file cut inside its last record (a BGZF EOF block behind it, so the container is whole): the GPU ingest gives it up
    import gzip
    data = gzip.open(src, "rb").read()
    d3 = tmp_path / "cut"
    d3.mkdir()
    with open(d3 / "whole.bam", "wb") as fh:
        z = bamio._Bgzf(fh)
        z.write(data[:60000]), z._flush(), z.write(data[60000:len(data) - 20])
        z.close()
    bamio.repack_bam(str(d3 / "whole.bam"), str(d3 / "rand.bam"), 20000, index=False)
    shutil.copy(src + ".bai", d3 / "rand.bam.bai")
    p = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "rand.bam"], cwd=d3, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_TIMING": "1", "HPN_NGPU": "1"})
    h = subprocess.run([os.path.join(BIN, "bam_sliding_count"), "-o", "host", "rand.bam"], cwd=d3, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_BAM_GPU": "0"})
    assert p.returncode == h.returncode and b"[hpn] host ingest" in p.stderr, p.stderr.decode()
    if p.returncode == 0:
        assert open(d3 / "out.txt", "rb").read() == open(d3 / "host.txt", "rb").read()


def test_bgzipped_fastq_is_inflated_on_the_gpu(tmp_path):
    """bgzip-style FASTQ: compressed blocks go to the GPU, are inflated there and framed where they land;
    the report equals the one for the same text read through zlib on the host."""
    from highperformancengs_amd.bamio import _Bgzf
    text = open(golden_path("fastq", "syn_var_a.fq"), "rb").read() * 30
    with open(tmp_path / "s.fq.gz", "wb") as fh:
        z = _Bgzf(fh)
        for i in range(0, len(text), 40000):
            z.write(text[i:i + 40000])
        z.close()
    outs = []
    # (HPN_BGZF_SLICE: one inflate launch can hold more text than one framing call takes -- 2 GiB -- so the text is cut into
    #  slices anywhere; 5000-byte slices put many cuts into this small file)
    # (HPN_NGPU: chunks of whole blocks to several lanes in turn, host/bgzf_shard.hpp -- the records that straddle chunks are framed
    #  with the byte before and the 4 KiB after handed over between the lanes)
    for env in ({}, {"HPN_BAM_CHUNK": "70000"}, {"HPN_NO_BGZF": "1", "HPN_NO_MGZ": "1", "HPN_TEXT": "0"}, {"HPN_BGZF_SLICE": "5000"},
                {"HPN_BGZF_SLICE": "65537", "HPN_BAM_CHUNK": "70000"}, {"HPN_NGPU": "2", "HPN_TIMING": "1"},
                {"HPN_NGPU": "3", "HPN_BAM_CHUNK": "70000", "HPN_TEXT_SLICE": "30000", "HPN_TIMING": "1"},
                {"HPN_NGPU": "8", "HPN_BAM_CHUNK": "140000", "HPN_TIMING": "1"}):
        p = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", "s.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env={**os.environ, **env})
        assert p.returncode == 0, p.stderr.decode()
        outs.append(p.stdout)
        if "HPN_NGPU" in env:
            assert f"one BGZF input over {env['HPN_NGPU']} lanes".encode() in p.stderr and b"abandoned" not in p.stderr, p.stderr.decode()
    assert all(o == outs[0] for o in outs)
    want = orc.fastq_count_report([str(tmp_path / "s.fq.gz")], names=["s.fq.gz"], header=True, length_detail=True)
    assert outs[0] == want
    # the same bytes with a damaged block in the middle: the GPU route gives up, zlib's verdict stands
    raw = bytearray(open(tmp_path / "s.fq.gz", "rb").read())
    raw[len(raw) // 2] ^= 0x55
    open(tmp_path / "bad.fq.gz", "wb").write(raw)
    a = subprocess.run([os.path.join(BIN, "fastq_count"), "bad.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([os.path.join(BIN, "fastq_count"), "bad.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_BAM_GPU": "0"})
    assert a.returncode == b.returncode and a.stdout == b.stdout
    c = subprocess.run([os.path.join(BIN, "fastq_count"), "bad.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env={**os.environ, "HPN_NGPU": "3", "HPN_BAM_CHUNK": "70000", "HPN_TIMING": "1"})
    assert c.returncode == a.returncode and c.stdout == a.stdout, c.stderr.decode()     # (a flipped byte inside a block's literals inflates "fine" on every route)
    # fastq_trim takes the same route when it writes to a file: identical text from every route; a damaged block
    # or irregular text (here: a read shorter than -s, the reference's stale-buffer case) makes it start over
    irregular = text + b"@short\nACG\n+\nIII\n" + text[:5000]
    with open(tmp_path / "irr.fq.gz", "wb") as fh:
        z = _Bgzf(fh)
        for i in range(0, len(irregular), 40000):
            z.write(irregular[i:i + 40000])
        z.close()
    for name in ("s.fq.gz", "irr.fq.gz", "bad.fq.gz"):
        outs = []
        for k, env in enumerate(({}, {"HPN_BAM_CHUNK": "70000"}, {"HPN_BAM_GPU": "0"},
                                 {"HPN_NO_BGZF": "1", "HPN_NO_MGZ": "1", "HPN_TEXT": "0"})):
            p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", name, "-o", f"o{k}", "-s", "5", "-e", "60"], cwd=tmp_path,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, **env})
            assert p.returncode == 0, p.stderr.decode()
            outs.append((open(tmp_path / f"o{k}.trim.fastq", "rb").read(), p.stderr.split(b"\n")[0]))
        if name == "bad.fq.gz":   # which bytes survive a damaged block depends on the reader; the two zlib-free GPU routes agree
            assert outs[0] == outs[1]
        else:
            assert outs[0] == outs[1] == outs[2] == outs[3], name
            assert outs[0][1].startswith(b"Total_reads: ") and len(outs[0][0]) > 100000


def test_single_member_gzip_is_inflated_on_the_gpu(tmp_path):
    """A plain .fastq.gz (one member, or several: cat a.gz b.gz): the host finds deflate block starts, the GPU inflates the
    stretches with the history unknown, resolves it and frames the text; the report equals zlib's route for several stretch
    sizes and batch splits.  Files the route must hand back: trailing bytes, damage, non-FASTQ text."""
    import gzip
    import zlib
    rng = np.random.default_rng(8)
    n = 30000
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 100))
    qual = rng.integers(35, 74, (n, 100), dtype=np.uint8)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, seq[i].tobytes(), qual[i].tobytes()) for i in range(n))   # 6.4 MB
    files = {"one.fq.gz": gzip.compress(text, 6), "lvl1.fq.gz": gzip.compress(text, 1),
             "two.fq.gz": gzip.compress(text[:len(text) // 2], 6) + gzip.compress(text[len(text) // 2:], 6),
             "many.fq.gz": b"".join(gzip.compress(text[a:a + 200_001], 1 + a % 9) for a in range(0, len(text), 200_001)),
             # members of one deflate block each: no block of theirs is a non-final one, the stretches start at member headers
             "tiny.fq.gz": b"".join(gzip.compress(text[a:a + 2143], 6) for a in range(0, len(text), 2143)),
             "tail.fq.gz": gzip.compress(text, 6) + b"trailing bytes\n",
             "ragged.fq.gz": gzip.compress(text + b"@x\nACGT\n+\nII\n" + text[:3000], 6)}
    bad = bytearray(files["one.fq.gz"])
    bad[len(bad) // 2] ^= 0x10
    files["bad.fq.gz"] = bytes(bad)
    for name, blob in files.items():
        (tmp_path / name).write_bytes(blob)
        ref = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", name], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             env={**os.environ, "HPN_NO_MGZ": "1", "HPN_NO_BGZF": "1"})
        # (HPN_GZ_FIND: the block starts looked for by the cores / by the device, k_gz_find_starts; several batches: the next
        # one is prepared by the producer thread while this one is inflated)
        for env in ({"HPN_GZ_GPU_FORCE": "1"}, {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "40000", "HPN_GZ_FIND": "host"},
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "150000", "HPN_GZ_BATCH": "7"},
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "40000", "HPN_GZ_FIND": "device"},
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "100000", "HPN_GZ_BATCH": "5", "HPN_GZ_FIND": "device"},
                    # a batch's text framed WHERE IT LIES in several slices (round 6: hpn_fastq_text_count_inplace; the carried bytes of a
                    # slice are laid over the tail of the slice before, of a batch's last slice in front of the next batch's text)
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "60000", "HPN_GZ_BATCH": "9", "HPN_TEXT_SLICE": "70001"},
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "150000", "HPN_GZ_BATCH": "3", "HPN_TEXT_SLICE": "4099"},
                    # the same file over several lanes (host/gz_shard.hpp): batches in turn, windows / member state / lines handed on
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_NGPU": "3", "HPN_GZ_STRETCH": "40000", "HPN_GZ_BATCH": "7", "HPN_TEXT_SLICE": "300000"},
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_NGPU": "2", "HPN_GZ_STRETCH": "150000", "HPN_GZ_BATCH": "4", "HPN_GZ_FIND": "device"},
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_NGPU": "5", "HPN_GZ_STRETCH": "60000", "HPN_GZ_BATCH": "6"}):
            p = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", name], cwd=tmp_path, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1", **env})
            assert p.returncode == ref.returncode, p.stderr.decode()
            if name != "bad.fq.gz":   # (what survives a damaged stream depends on the reader)
                assert p.stdout == ref.stdout, (name, env)
            used = b"[hpn] gzip on the GPU" in p.stderr or b"one gzip input over" in p.stderr
            assert used == (name in ("one.fq.gz", "lvl1.fq.gz", "two.fq.gz", "many.fq.gz", "tiny.fq.gz")), (name, env, p.stderr)
            if "HPN_NGPU" in env and used:
                assert f"one gzip input over {env['HPN_NGPU']} lanes".encode() in p.stderr, (name, env, p.stderr)
    # fastq_trim to a file takes the same route; whatever it has to hand back (here also: a read shorter than -s) starts over
    for name in ("one.fq.gz", "two.fq.gz", "many.fq.gz", "ragged.fq.gz", "tail.fq.gz"):
        outs = []
        for k, env in enumerate(({"HPN_GZ_GPU_FORCE": "1"}, {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "90000", "HPN_GZ_BATCH": "11"},
                                 {"HPN_GZ_GPU": "0"}, {"HPN_NO_MGZ": "1", "HPN_TEXT": "0"})):
            p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", name, "-o", f"t{k}", "-s", "5", "-e", "60"], cwd=tmp_path,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "HPN_TIMING": "1", **env})
            assert p.returncode == 0, p.stderr.decode()
            outs.append((open(tmp_path / f"t{k}.trim.fastq", "rb").read(), [l for l in p.stderr.split(b"\n") if l.startswith(b"Total_reads")]))
            if k < 2:
                assert (b"[hpn] gzip on the GPU" in p.stderr) == (name in ("one.fq.gz", "two.fq.gz", "many.fq.gz")), (name, p.stderr)
        assert outs[0] == outs[1] == outs[2] == outs[3], name
    want = orc.fastq_count_report([str(tmp_path / "one.fq.gz")], names=["one.fq.gz"], header=True, length_detail=True)
    assert ref is not None and subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", "one.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE,
                                              env={**os.environ, "HPN_GZ_GPU_FORCE": "1"}).stdout == want


def test_fastq_trim_reports_total_reads(tmp_path):
    p, _ = _run("fastq_trim", ["-i", "t.fq", "-s", "2", "-e", "8"], [golden_path("fastq", "t.fq")], tmp_path)
    assert p.stderr.startswith(b"Total_reads: 5\nFinished in ")


def test_bam2depth_requires_index(tmp_path):
    shutil.copy(golden_path("bam", "e.bam"), tmp_path)
    p = subprocess.run([os.path.join(BIN, "bam2depth"), "-o", "d", "e.bam"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"BAM indexing file is not available" in p.stderr


@pytest.mark.parametrize("case,why", [("count_badcrc", b"CRC-32 mismatch"), ("count_badcrc_mid", b"CRC-32 mismatch in a member"),
                                      ("count_badisize", b"ISIZE mismatch")])
def test_gzip_on_the_gpu_checks_every_member_like_gzread(manifest, case, why, tmp_path):
    """The device inflate route takes the CRC-32 of each member's text (hpn_crc32_dev) and compares it and ISIZE with the
    trailer, as zlib's gzread does behind the reference's gzgets; a member that fails sends the file to zlib's own reader,
    whose bytes -- up to the buffer the failure was noticed in -- are the reference's (goldens made by the reference binary)."""
    c = manifest[case]
    for env in ({"HPN_GZ_GPU_FORCE": "1"}, {"HPN_GZ_GPU_FORCE": "1", "HPN_GZ_STRETCH": "40000"}):
        p, files = _run(c["tool"], ["-t", "1"] + list(c["args"]), [os.path.join(GOLDEN, i) for i in c["inputs"]], tmp_path, {**env, "HPN_TIMING": "1"})
        assert p.returncode == 0 and p.stdout == expected(case), p.stderr.decode()
        assert b"gzip route on the GPU abandoned" in p.stderr and why in p.stderr, p.stderr.decode()
    # and a sound file passes the same checks
    c = manifest["count_multi"]
    p, files = _run(c["tool"], ["-t", "1"] + list(c["args"]), [os.path.join(GOLDEN, i) for i in c["inputs"]], tmp_path,
                    {"HPN_GZ_GPU_FORCE": "1", "HPN_TIMING": "1"})
    assert p.stdout == expected("count_multi") and b"gzip on the GPU: inflate + frame + tally" in p.stderr, p.stderr.decode()


def _reads_file(path, n, L, seed):
    seq, qual, off = orc.synth_soa(seed, 0, n, L, L)
    s, q = seq.reshape(n, L), qual.reshape(n, L)
    with open(path, "wb") as fh:
        for i in range(n):
            fh.write(b"@read%d/1\n" % i + s[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")


def test_trim_appending_to_a_file_that_has_content(tmp_path):
    """`fastq_trim ... -o - >> all.fq`: stdout is a regular file opened O_APPEND and the slabs are megabytes.  Round 4's writer
    cut such slabs into pieces for several pwrite threads, and pwrite ignores its offset on an O_APPEND descriptor: the pieces
    landed in arrival order (round-4 advisor).  The writer is one thread with plain write() now (host/text_stream.hpp:
    write_slab): what was there stays, what follows is the run's output byte for byte.  Reference: fprintf in order, fastq_trim.c:101."""
    n, L = 60000, 150                                    # ~19 MB in, ~17 MB out: several slabs of more than 4 MiB
    _reads_file(tmp_path / "r.fq", n, L, 4242)
    p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", "r.fq", "-s", "5", "-e", "140", "-o", "fresh"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    want = open(tmp_path / "fresh.trim.fastq", "rb").read()
    assert len(want) > (12 << 20)
    head = b"@kept\nACGT\n+\nIIII\n" * 1000
    with open(tmp_path / "all.fq", "wb") as f:
        f.write(head)
    for env in ({}, {"HPN_NGPU": "2"}):
        with open(tmp_path / "all.fq", "ab") as f:
            p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", "r.fq", "-s", "5", "-e", "140", "-o", "-"], cwd=tmp_path, stdout=f, stderr=subprocess.PIPE,
                               env={**os.environ, **env})
        assert p.returncode == 0, p.stderr.decode()
        head += want
        assert open(tmp_path / "all.fq", "rb").read() == head, env


def test_a_write_that_fails_ends_the_tool_with_a_code(tmp_path):
    """/dev/full takes no byte (ENOSPC).  The reference's fprintf never looks and exits 0 with a short output; here a write that
    does not go through ends fastq_trim with exit code 2 and a line on stderr (round-4 advisor: it printed a message and exited 0)."""
    _reads_file(tmp_path / "r.fq", 20000, 150, 77)
    with open("/dev/full", "wb") as f:
        p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", "r.fq", "-s", "5", "-e", "140", "-o", "-"], cwd=tmp_path, stdout=f, stderr=subprocess.PIPE)
    assert p.returncode == 2 and b"writing the output failed" in p.stderr, (p.returncode, p.stderr.decode()[-500:])


def test_feeders_next_to_the_device_or_not_same_bytes(manifest, tmp_path):
    """HPN_NUMA=0 leaves the reader / uploader threads where the scheduler puts them (host/cpus.hpp: bind_thread_near); the outputs
    cannot depend on it.  Two golden cases per tool family on both settings."""
    for case in ("count_syn_100", "trim_syn_100", "depth_rand", "sliding_rand"):
        for k, env in enumerate(({"HPN_NUMA": "0"}, {"HPN_NUMA": "1"})):
            d = tmp_path / f"{case}_{k}"
            d.mkdir()
            _check(manifest, case, d, env)


@pytest.mark.parametrize("tool,threads", [("fastq_count", "1"), ("fastq_count", "4"), ("fastq_count_kthread", "4")])
def test_a_read_of_512_bases_ends_the_tool_with_code_2_whatever_the_other_workers_do(tool, threads, tmp_path):
    """SeqLen[512] has no slot for it (the reference writes behind the array): message, exit code 2, no report.  With other
    workers still inside the runtime the way out must not run the runtime's exit handlers -- fastq_count_kthread -t 4 died of
    SIGSEGV here instead (scripts/soak_fastq_tools.py, round 6)."""
    rng = np.random.default_rng(3)
    names = []
    for k in range(4):
        n = 20000
        recs = [b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 100)), bytes(rng.integers(35, 74, 100, dtype=np.uint8)))
                for i in range(n)]
        if k == 2:
            recs[n // 2] = b"@long\n" + b"A" * 600 + b"\n+\n" + b"I" * 600 + b"\n"
        names.append("f%d.fq" % k)
        (tmp_path / names[-1]).write_bytes(b"".join(recs))
    p = subprocess.run([os.path.join(BIN, tool), "-t", threads, "-o", "m.tsv"] + names, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 2, (p.returncode, p.stderr.decode())
    assert b"read longer than 511 bases" in p.stderr


@pytest.mark.parametrize("damage", ["literal", "isize"])
def test_bgzip_fastq_with_a_block_that_fails_its_check_ends_where_gzread_ends(damage, tmp_path):
    """A bgzip-compressed FASTQ is a gzip file to the reference: gzread checks every member's CRC-32 and ISIZE and hands out
    nothing from the failing member on.  samtools' BGZF reader (what the BAM tools stand in for) checks no CRC, and until round 6
    neither did the text routes built on the same decoders: a flipped bit inside a literal passed through fastq_count and
    fastq_trim on the device and on the host alike (found by scripts/soak_fastq_tools.py)."""
    import struct
    import zlib
    rng = np.random.default_rng(17)
    n = 9000
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 100)), bytes(rng.integers(35, 74, 100, dtype=np.uint8)))
                    for i in range(n))
    blocks = []
    for a in range(0, len(text), 30000):
        piece = text[a:a + 30000]
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)      # (literals only: a flipped bit changes a byte, not the length)
        comp = co.compress(piece) + co.flush()
        blocks.append(bytearray(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                                struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece))))
    k = len(blocks) // 2
    if damage == "literal":
        for at in range(len(blocks[k]) // 2, len(blocks[k]) - 8):               # a bit that leaves the block decodable to its stated length
            trial = bytearray(blocks[k])
            trial[at] ^= 4
            try:
                if len(zlib.decompress(bytes(trial[18:-8]), -15)) == 30000:
                    blocks[k] = trial
                    break
            except zlib.error:
                continue
        else:
            pytest.skip("no length-preserving bit found")
    else:
        blocks[k][-4:] = struct.pack("<I", 29999)
    blob = b"".join(bytes(b) for b in blocks) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    (tmp_path / "d.fq.gz").write_bytes(blob)
    want = orc.fastq_count_report([str(tmp_path / "d.fq.gz")], names=["d.fq.gz"], header=True, length_detail=True)
    assert int(want.split(b"\n")[1].split(b"\t")[1]) < n          # the reference stops at the member
    for env in ({}, {"HPN_BAM_GPU": "0"}, {"HPN_NGPU": "2"}):
        p = subprocess.run([os.path.join(BIN, "fastq_count"), "-H", "-L", "d.fq.gz"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env={**os.environ, **env})
        assert p.returncode == 0 and p.stdout == want, (env, p.stderr.decode())
        p = subprocess.run([os.path.join(BIN, "fastq_trim"), "-i", "d.fq.gz", "-o", "t", "-s", "0", "-e", "60"], cwd=tmp_path, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env={**os.environ, **env})
        assert p.returncode == 0, (env, p.stderr.decode())
        got = open(tmp_path / "t.trim.fastq", "rb").read()
        reads = int(want.split(b"\n")[1].split(b"\t")[1])
        assert reads - 1 <= got.count(b"\n+\n") <= reads, (env, got.count(b"\n+\n"), reads)     # (the record the failure falls into: test_trim_of_a_damaged_gzip)
        os.unlink(tmp_path / "t.trim.fastq")
