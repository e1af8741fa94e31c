"""CPU: the host side of "one FASTQ over several lanes" (csrc/host/text_shard.hpp -- reader, dispatcher, lane threads, the board of
line counts, the ordered writer) with the few ABI calls it makes stood in for by tests/stub/shard_harness.cpp: the pieces' results
must add up to a serial pass, for any number of lanes and piece size, under ThreadSanitizer too (scripts/sanitize_shard.sh runs the
full matrix under TSAN and ASAN + UBSan).  The device halves of the piece API are tested in tests/test_text_piece_gpu.py."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("shard")
    out = {}
    for tag, flags in (("plain", ["-O1"]), ("tsan", ["-O1", "-g", "-fsanitize=thread"])):
        exe = str(d / tag)
        subprocess.check_call(["g++", "-std=c++17", "-DHPN_TEST_HOOKS", *flags, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "stub", "shard_harness.cpp"),
                               "-o", exe, "-lz", "-lpthread"])
        out[tag] = exe
    return out


def _text(seed, n):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        l = int(rng.integers(0, 180))
        recs.append((b"@r%d %s" % (i, bytes(rng.integers(48, 123, int(rng.integers(0, 25)), dtype=np.uint8))),
                     bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), l)), bytes(rng.integers(33, 75, l, dtype=np.uint8))))
    return recs


@pytest.mark.parametrize("lanes,chunk", [(2, 8192), (3, 20000), (5, 8192), (4, 1 << 20)])
def test_pieces_add_up_to_the_serial_pass(harness, tmp_path, lanes, chunk):
    recs = _text(lanes * 7 + chunk % 13, 4000)
    text = b"".join(b"%s\n%s\n+\n%s\n" % r for r in recs)
    (tmp_path / "a.fq").write_bytes(text)
    (tmp_path / "a.fq.gz").write_bytes(b"".join(gzip.compress(text[i:i + 200000], 6) for i in range(0, len(text), 200000)))
    want = "0 0 %d %d" % (len(recs), sum(len(r[1]) for r in recs))
    env = {**os.environ, "HPN_TEXT_CHUNK": str(chunk)}
    for f in ("a.fq", "a.fq.gz"):
        p = subprocess.run([harness["plain"], "count", str(tmp_path / f), str(lanes)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.stdout.decode().strip() == want, (f, p.stderr.decode())
    S, E = 4, 77
    p = subprocess.run([harness["plain"], "trim", str(tmp_path / "a.fq"), str(lanes), str(S), str(E), str(tmp_path / "t.out")], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env)
    assert p.stdout.decode().strip() == "0 0 %d" % len(recs)
    assert (tmp_path / "t.out").read_bytes() == b"".join(b"%s\n%s\n+\n%s\n" % (n, s[S:E], q[S:E]) for n, s, q in recs)
    # a stream that ends inside a record's '+' line: the route is abandoned, nothing is added
    half = b"".join(b"%s\n%s\n+\n%s\n" % r for r in recs[:2000])
    (tmp_path / "trunc.fq").write_bytes(half + b"%s\n%s\n+" % recs[2000][:2])
    p = subprocess.run([harness["plain"], "count", str(tmp_path / "trunc.fq"), str(lanes)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.stdout.decode().strip() == "0 1 0 0"


def test_no_data_race_between_reader_dispatcher_and_lanes(harness, tmp_path):
    recs = _text(99, 3000)
    text = b"".join(b"%s\n%s\n+\n%s\n" % r for r in recs)
    (tmp_path / "a.fq").write_bytes(text)
    env = {**os.environ, "HPN_TEXT_CHUNK": "8192", "TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0"}
    for args in (["count", str(tmp_path / "a.fq"), "4"], ["trim", str(tmp_path / "a.fq"), "3", "2", "60", str(tmp_path / "t.out")]):
        p = subprocess.run([harness["tsan"]] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0 and b"ThreadSanitizer" not in p.stderr, p.stderr.decode()[:2000]
        assert p.stdout.decode().startswith("0 0 %d" % len(recs))


@pytest.mark.parametrize("lanes", [2, 3])
@pytest.mark.parametrize("sleep_ms", [0, 300])
def test_trim_with_an_irregular_piece_in_the_middle_falls_back_instead_of_hanging(harness, tmp_path, lanes, sleep_ms):
    """One 100 kB record in the middle (longer than a piece's 4 KiB tail) makes its piece irregular while the other lanes already hold
    both of their output slabs queued behind its sequence number: PieceRun::stop() must wake OrderedWriter::acquire() (round-3 advisor:
    hung 1 in 40, always with the irregular lane slowed).  Expected: the route reports "irregular" and returns; never a timeout."""
    recs = _text(5, 3000)
    big = (b"@big", b"A" * 100_000, b"I" * 100_000)
    text = b"".join(b"%s\n%s\n+\n%s\n" % r for r in recs[:1500] + [big] + recs[1500:])
    (tmp_path / "b.fq").write_bytes(text)
    env = {**os.environ, "HPN_TEXT_CHUNK": "65536", "TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0"}
    if sleep_ms:
        env["HPN_STUB_IRREGULAR_SLEEP_MS"] = str(sleep_ms)
    for tag, reps in (("plain", 10 if not sleep_ms else 2), ("tsan", 1)):
        for _ in range(reps):
            p = subprocess.run([harness[tag], "trim", str(tmp_path / "b.fq"), str(lanes), "3", "90", str(tmp_path / "t.out")], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, env=env, timeout=60)
            assert p.stdout.decode().split()[:2] == ["0", "1"], (p.stdout, p.stderr.decode()[:1000])
            assert b"ThreadSanitizer" not in p.stderr
