"""The R plugin's per-read tally (Rgzfastq_uniq.c STATSEQ / AssignQuality / Length).

PARITY UNPINNED: R's headers are not in the image, so the plugin cannot be built and the
reference's tests hold no vector for it.  The oracle restates the three macros from the source
text; here it is checked by hand on SURVEY Appendix A.1's five reads, and the GPU path is held
to it bit for bit (counts) / to the last ulp (the double quotient is one IEEE division)."""
import numpy as np
import pytest

import orc
from conftest import golden_path

SEQS = [b"NAGATTTTCA", b"GAAANATCTA", b"ATNACGAGNTNC", b"CGNGATNACNTGTAT", b"NGNGTGNNATNC"]
QUALS = [b'@"9<G!=2/F', b"B/=@D/7//>", b"43F@A:F#?0:;", b"#4HFF:++A/!-CD/", b"BD.<$?8ED-A;"]


def _soa():
    seq = np.frombuffer(b"".join(SEQS), np.uint8)
    qual = np.frombuffer(b"".join(QUALS), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(s) for s in SEQS])]).astype(np.uint64)
    return seq, qual, off


def test_oracle_by_hand_on_appendix_a1():
    rc, r = orc.rqc_soa(*_soa())
    assert rc == 0
    # Length[len-1]
    want_len = np.zeros(300, np.int32)
    for s in SEQS:
        want_len[len(s) - 1] += 1
    assert np.array_equal(r["length"], want_len)
    # Nucleotide[5*pos+code], codes T0 C1 A2 G3 N4
    code = {ord("T"): 0, ord("C"): 1, ord("A"): 2, ord("G"): 3, ord("N"): 4}
    want_nuc = np.zeros((300, 5), np.int32)
    want_q = np.zeros((300, 128), np.int32)
    for s, q in zip(SEQS, QUALS):
        for i, b in enumerate(s):
            want_nuc[i, code[b]] += 1
        for i, b in enumerate(q):
            want_q[i, b] += 1
    assert np.array_equal(r["nucleotide"], want_nuc) and np.array_equal(r["quality"], want_q)
    assert r["nucleotide"][0].tolist() == [0, 1, 1, 1, 2]  # cycle 1: N G A C N
    want_gc = [sum(b in b"GC" for b in s) / len(s) for s in SEQS]
    assert r["gc"].tolist() == want_gc
    assert r["gc"][0] == 2 / 10  # NAGATTTTCA


def test_oracle_stream_equals_soa_and_lowercase_rules(tmp_path):
    rc, a = orc.rqc_stream(golden_path("fastq", "t.fq"))
    rc2, b = orc.rqc_soa(*_soa())
    assert rc == 0 and rc2 == 0
    for k in ("quality", "nucleotide", "length", "gc"):
        assert np.array_equal(a[k], b[k]), k
    # initNtVal: lower case acgt map like upper case, 'n' does NOT map to N, '.' does; GC counts upper case only
    seq = np.frombuffer(b"acgtnN.xGc", np.uint8)
    rc, r = orc.rqc_soa(seq, np.full(10, 40, np.uint8), np.array([0, 10], np.uint64))
    assert rc == 0
    assert [int(np.argmax(r["nucleotide"][i])) for i in range(10)] == [2, 1, 3, 0, 0, 4, 4, 0, 3, 1]
    assert r["gc"][0] == 1 / 10
    # domain: empty read (Length[-1]), read longer than MaxLen
    assert orc.rqc_soa(seq, seq, np.array([0, 0], np.uint64))[0] != 0
    long = np.full(301, 65, np.uint8)
    assert orc.rqc_soa(long, long, np.array([0, 301], np.uint64))[0] != 0


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    import highperformancengs_amd as hp
    c = hp.Context(0)
    yield c
    c.close()


def _same(got, want):
    for k in ("quality", "nucleotide", "length"):
        assert np.array_equal(got[k], want[k]), k
    assert got["gc"].tobytes() == want["gc"].tobytes()  # bit-identical doubles (NaN-safe)


@pytest.mark.gpu
def test_gpu_appendix_a1(ctx):
    rc, want = orc.rqc_soa(*_soa())
    _same(ctx.fastq_rqc(*_soa()), want)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,lo,hi", [(1, 1, 1, 1), (2, 100, 1, 15), (3, 1000, 16, 16), (4, 5000, 1, 300), (5, 4097, 150, 150),
                                          (6, 3000, 17, 47), (7, 200000, 100, 100), (8, 64, 300, 300)])
def test_gpu_synthetic(ctx, seed, n, lo, hi):
    seq, qual, off = orc.synth_soa(100 + seed, seed * 1000, n, lo, hi)
    if seed == 4:  # lower case, '.', stray bytes
        seq = seq.copy()
        rng = np.random.default_rng(seed)
        idx = rng.integers(0, len(seq), len(seq) // 7)
        seq[idx] = rng.choice(np.frombuffer(b"acgtnu.UxG", np.uint8), len(idx))
    rc, want = orc.rqc_soa(seq, qual, off)
    assert rc == 0
    _same(ctx.fastq_rqc(seq, qual, off), want)


@pytest.mark.gpu
def test_gpu_accumulates_and_sub_batches(ctx):
    seq, qual, off = orc.synth_soa(9, 0, 3000, 20, 120)
    rc, want = orc.rqc_soa(seq, qual, off)
    out = ctx.fastq_rqc(seq, qual, off[:1001])
    gc = [out["gc"]]
    out = ctx.fastq_rqc(seq, qual, off[1000:], out=out)  # offsets not starting at 0
    gc.append(out["gc"])
    out["gc"] = np.concatenate(gc)
    _same(out, want)


@pytest.mark.gpu
def test_gpu_domain(ctx):
    from highperformancengs_amd import HpnError
    a = np.full(301, 65, np.uint8)
    with pytest.raises(HpnError):
        ctx.fastq_rqc(a, a, np.array([0, 301], np.uint64))
    with pytest.raises(HpnError):
        ctx.fastq_rqc(a, a, np.array([0, 0, 5], np.uint64))
    rc, want = orc.rqc_soa(a[:300], a[:300], np.array([0, 300], np.uint64))
    _same(ctx.fastq_rqc(a[:300], a[:300], np.array([0, 300], np.uint64)), want)


# ---- half of the pin that needs no R: the plugin's Quality[q + 128 pos] and Length[len - 1] (Rgzfastq_uniq.c:42-48,174)
# are the transposes of what the reference's fastq_count_kthread -L prints for the same reads (fastq_count_kthread.c:52-64),
# and those bytes are golden (tests/golden/expected/kthread_*).  Nucleotide and the per-read GC stay unpinned.

def _read_fastq(path):
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        lines = f.read().split(b"\n")
    seqs, quals = lines[1::4], lines[3::4]
    seqs, quals = seqs[:len(quals)], quals[:len(seqs)]
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return np.frombuffer(b"".join(seqs), np.uint8), np.frombuffer(b"".join(quals), np.uint8), off


def _reference_matrix(case, name, idx):
    from conftest import expected
    lines = [l for l in expected(case, f"{name}.{idx}.tsv").decode().split("\n") if l and not l.startswith("#Filename")]
    lens = [int(x) for x in lines[1].split("\t")[1:]]
    freq = [int(x) for x in lines[2].split("\t")[1:]]
    mat = np.array([[int(x) for x in l.split("\t")] for l in lines[3:3 + 128]], np.int64)   # [quality byte][cycle]
    return lens, freq, mat


HALF_PIN = [("kthread_a1", "t.fq", 0), ("kthread_a1", "t.fq.gz", 1), ("kthread_syn", "syn_var_a.fq", 0),
            ("kthread_syn", "syn_var_b.fq.gz", 1), ("kthread_syn", "syn_100.fq.gz", 2)]


def _check_half_pin(r, case, name, idx):
    lens, freq, mat = _reference_matrix(case, name, idx)
    max_len = mat.shape[1]
    assert np.array_equal(r["quality"][:max_len].T.astype(np.int64), mat) and int(r["quality"][max_len:].sum()) == 0
    want_len = np.zeros(300, np.int64)
    for l, f in zip(lens, freq):
        want_len[l - 1] = f
    assert np.array_equal(r["length"].astype(np.int64), want_len)


@pytest.mark.parametrize("case,name,idx", HALF_PIN)
def test_oracle_quality_and_length_equal_the_reference_kthread_matrix(case, name, idx):
    rc, r = orc.rqc_soa(*_read_fastq(golden_path("fastq", name)))
    assert rc == 0
    _check_half_pin(r, case, name, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("case,name,idx", HALF_PIN)
def test_gpu_quality_and_length_equal_the_reference_kthread_matrix(ctx, case, name, idx):
    _check_half_pin(ctx.fastq_rqc(*_read_fastq(golden_path("fastq", name))), case, name, idx)
