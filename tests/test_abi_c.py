"""include/hpngs.h is a C header and the bindings INTEGRATION.md shows are real: tests/abi/*.c are the Seam 1
(count_read) and Seam 3 (bam_fetch_f callback) stubs as C99 programs, compiled with -std=c99 -pedantic -Wall -Werror
against the header and linked with libhpngs only.  CPU: they compile and link.  GPU: their numbers equal the
reference tools' own outputs (tests/golden/expected)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import expected, golden_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "highperformancengs_amd")


def _cc(src, out, extra=()):
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi", src), "-o", str(out), "-L" + LIBDIR, "-lhpngs", "-lz",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", *extra]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()
    return str(out)


def test_header_is_plain_c(tmp_path):
    (tmp_path / "h.c").write_text('#include "hpngs.h"\nint main(void) { return (int)sizeof(hpn_tally) == 0; }\n')
    for std in ("c99", "c11"):
        p = subprocess.run(["gcc", f"-std={std}", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                            "-I" + os.path.join(ROOT, "include"), str(tmp_path / "h.c")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert p.returncode == 0, p.stdout.decode()
    # and from C++ (the tools)
    (tmp_path / "h.cpp").write_text('#include "hpngs.h"\nint main() { return 0; }\n')
    p = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                        str(tmp_path / "h.cpp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()


@pytest.mark.parametrize("src", ["seam1_count_read.c", "seam3_fetch_func.c", "comm_one_rank.c"])
def test_integration_stubs_compile_and_link(src, tmp_path):
    exe = _cc(src, tmp_path / src[:-2])
    # every hpn_* symbol the program needs comes from libhpngs.so
    out = subprocess.run(["ldd", exe], stdout=subprocess.PIPE).stdout.decode()
    assert "libhpngs.so" in out


def _kthread_tsv(case, name, idx):
    """numbers of the reference's per-file report <name>.<idx>.tsv with -L: row, #Freq row, 128 x maxLen matrix"""
    lines = expected(case, f"{name}.{idx}.tsv").decode().split("\n")
    lines = [l for l in lines if l and not l.startswith("#Filename")]
    row = lines[0].split("\t")
    lens = [int(x) for x in lines[1].split("\t")[1:]]
    freq = [int(x) for x in lines[2].split("\t")[1:]]
    mat = np.array([[int(x) for x in l.split("\t")] for l in lines[3:3 + 128]], np.uint64)
    return row, lens, freq, mat


@pytest.mark.gpu
@pytest.mark.parametrize("case,name,idx", [("kthread_a1", "t.fq", 0), ("kthread_a1", "t.fq.gz", 1), ("kthread_syn", "syn_var_a.fq", 0),
                                           ("kthread_syn", "syn_var_b.fq.gz", 1), ("kthread_syn", "syn_100.fq.gz", 2)])
def test_seam1_count_read_matches_the_reference_report(case, name, idx, tmp_path):
    exe = _cc("seam1_count_read.c", tmp_path / "seam1")
    p = subprocess.run([exe, golden_path("fastq", name)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    out = p.stdout.decode().split("\n")
    reads, min_len, max_len, total, q20, q30 = (int(x) for x in out[0].split())
    seqlen = np.array(out[1].split(), np.uint64)
    quality = np.array([l.split() for l in out[2:130]], np.uint64)
    row, lens, freq, mat = _kthread_tsv(case, name, idx)
    assert row[1] == str(reads) and int(row[4]) == min_len and int(row[5]) == max_len
    assert row[2] == "%.0f" % float(int((seqlen * np.arange(512, dtype=np.uint64)).sum()))
    assert row[6] == "%.3f" % (1.0 * q20 / total * 100) and row[7] == "%.3f" % (1.0 * q30 / total * 100)
    assert lens == list(range(min_len, max_len + 1)) and freq == [int(seqlen[l]) for l in lens]
    assert mat.shape[1] == max_len and np.array_equal(quality[:, :max_len], mat) and int(quality[:, max_len:].sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case,bam,W,pre", [("depth_a3", "e.bam", 100, "d"), ("depth_rand", "rand.bam", 20000, "r"),
                                            ("depth_rand_w1000", "rand.bam", 1000, "r")])
def test_seam3_fetch_func_matches_the_reference_bedgraph(case, bam, W, pre, tmp_path):
    exe = _cc("seam3_fetch_func.c", tmp_path / "seam3")
    p = subprocess.run([exe, golden_path("bam", bam), str(W)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    lines = p.stdout.split(b"\n")
    bed = b"".join(l + b"\n" for l in lines if l and not l.startswith(b"#win"))
    assert bed == expected(case, f"{bam}.1.bedGraph")
    # the window sums give the reference's .depth rows (output_bins, bam2depth.c:238-246: bins[k] / W with %.2f)
    want = expected(case, f"{pre}.1.depth").split(b"\n")
    got = [l.split() for l in lines if l.startswith(b"#win")]
    assert len(got) == len([w for w in want if w])
    for g, w in zip(got, want):
        assert w.split(b"\t")[3] == ("%.2f" % (float(int(g[3])) / W)).encode()


@pytest.mark.gpu
def test_a_process_without_torch_resolves_the_real_rccl(tmp_path):
    """The C tools carry no torch: libhpngs must find librccl.so.1 on its own (dlopen), make communicators through both entry
    points and say which file carried the sum -- the first 8-GPU run must not be the first time this is tried."""
    exe = _cc("comm_one_rank.c", tmp_path / "comm_one_rank")
    env = {k: v for k, v in os.environ.items() if not k.startswith("HPN_")}
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()
    path = p.stdout.decode().strip().splitlines()[-1]            # (RCCL may print its version banner on stdout first)
    assert os.path.basename(path).startswith("librccl.so") and os.path.isfile(path), path
    assert "stub" not in path and "torch" not in path, path        # the system's library, not the test stand-in, not torch's copy
    # ... and nothing of torch was in that process
    assert b"torch" not in subprocess.run(["ldd", exe], stdout=subprocess.PIPE).stdout
