"""CPU: the shipped library and tools read USER knobs only (highperformancengs_amd/csrc/host/knobs.hpp).
Every test / timing switch -- HPN_RCCL_LIB, HPN_COMM_SHARED_DEVICE, HPN_TRIM_NOWRITE, the forced routes and chunk sizes -- is read
through test_env(), which is getenv() only under -DHPN_TEST_HOOKS: the names must not even occur in what ships, they must occur
in the hooks build, and tests/conftest.py's two lists must be exactly the names the sources read.  (no
reference counterpart: the reference reads no environment.)"""
import glob
import os
import re
import subprocess

from conftest import BIN, HOOKS_BIN, HOOKS_LIB, PKG, TEST_KNOBS, USER_KNOBS

CSRC = os.path.join(PKG, "csrc")
NOT_ENV = {"HPN_DEPTH_ANY_ORDER", "HPN_TALLY_NUC_HIST", "HPN_TALLY_WORDS", "HPN_TALLY_QUAL_HIST", "HPN_TEXT_PIECE_TAIL", "HPN_TEST_HOOKS"}   # constants of hpngs.h, a macro


def _names_in(path):
    out = subprocess.run(["strings", "-n", "5", path], stdout=subprocess.PIPE).stdout.decode()
    return set(re.findall(r"\bHPN_[A-Z0-9_]+\b", out)) - NOT_ENV


def _sources():
    return [f for ext in ("hpp", "hip", "cpp") for f in glob.glob(os.path.join(CSRC, "**", "*." + ext), recursive=True)]


def test_the_lists_are_what_the_sources_read():
    user, test = set(), set()
    for f in _sources():
        s = open(f).read()
        user |= set(re.findall(r'\bgetenv\("(HPN_[A-Z0-9_]+)"\)', s))
        test |= set(re.findall(r'\btest_env\("(HPN_[A-Z0-9_]+)"\)', s))
    assert user == USER_KNOBS, (sorted(user - USER_KNOBS), sorted(USER_KNOBS - user))
    assert test == TEST_KNOBS, (sorted(test - TEST_KNOBS), sorted(TEST_KNOBS - test))
    assert not (user & test)


def test_what_ships_knows_no_test_switch():
    shipped = [os.path.join(PKG, "libhpngs.so")] + sorted(glob.glob(os.path.join(BIN, "*")))
    assert len(shipped) >= 7
    for path in shipped:
        names = _names_in(path)
        assert not (names & TEST_KNOBS), (path, sorted(names & TEST_KNOBS))
        assert names <= USER_KNOBS, (path, sorted(names - USER_KNOBS))


def test_the_hooks_build_has_them():
    seen = _names_in(HOOKS_LIB)
    for path in glob.glob(os.path.join(HOOKS_BIN, "*")):
        seen |= _names_in(path)
    assert TEST_KNOBS <= seen, sorted(TEST_KNOBS - seen)


def test_the_shipped_build_names_variables_it_does_not_read(tmp_path):
    """A test switch set against the shipped binaries measures the default and calls it a sweep: the shipped build says
    once per process which HPN_* variables it does not read -- without naming any test switch in its own strings -- and stays quiet
    about the user knobs; the hooks build reads them and says nothing.  (Runs without a GPU: the line comes before the device is
    looked for.)"""
    (tmp_path / "x.fq").write_bytes(b"@r\nACGT\n+\nIIII\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("HPN_")}
    env.update({"HPN_GZ_STRETCH": "4096", "HPN_TIMING": "1", "HPN_NOT_A_KNOB": "1"})
    # (through a shell: tests/conftest.py hands a child whose environment holds a test switch the hooks build -- here the shipped
    # one is the point)
    p = subprocess.run(["/bin/sh", "-c", "exec " + os.path.join(BIN, "fastq_count") + " x.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    err = p.stderr.decode()
    assert "HPN_GZ_STRETCH is set but this build does not read it" in err and "HPN_NOT_A_KNOB is set" in err, err
    assert "HPN_TIMING is set" not in err
    assert err.count("HPN_GZ_STRETCH is set") == 1
    q = subprocess.run([os.path.join(HOOKS_BIN, "fastq_count"), "x.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert "does not read it" not in q.stderr.decode()
