import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# ---- the two builds (highperformancengs_amd/csrc/host/knobs.hpp) -----------------------------------------------------------
# The shipped library and tools read USER knobs only.  Test / timing switches (routes forced on small inputs, chunk sizes that
# cut test files into pieces, the RCCL stand-in, lanes sharing a device) exist only in the -DHPN_TEST_HOOKS build under
# highperformancengs_amd/testhooks/ -- the same sources.  A child process started with one of those switches in its environment
# is given that build: a tool under .../bin/ is swapped for its twin under .../testhooks/bin/, any other child (python -c ...)
# gets HPN_LIB = the hooks library.  Children without such a switch run what ships.
PKG = os.path.join(ROOT, "highperformancengs_amd")
BIN, HOOKS_BIN, HOOKS_LIB = os.path.join(PKG, "bin"), os.path.join(PKG, "testhooks", "bin"), os.path.join(PKG, "testhooks", "libhpngs.so")
USER_KNOBS = {"HPN_DEVICE", "HPN_NGPU", "HPN_TIMING", "HPN_FULL_EXIT", "HPN_NUMA", "HPN_READ_THREADS", "HPN_GZ_THREADS", "HPN_BGZF_THREADS",
              "HPN_TEXT", "HPN_BAM_GPU", "HPN_GZ_GPU", "HPN_BEDGRAPH_HOST", "HPN_DEPTH_LOOKAHEAD", "HPN_ALLREDUCE"}
TEST_KNOBS = {"HPN_RCCL_LIB", "HPN_COMM_SHARED_DEVICE", "HPN_TRIM_NOWRITE", "HPN_ALL_WORKERS", "HPN_BAM_AHEAD", "HPN_BAM_CHUNK", "HPN_BAM_ROUNDS",
              "HPN_BGZF_SLICE", "HPN_FAST_INFLATE", "HPN_GZ_BATCH", "HPN_GZ_CRC", "HPN_GZ_DEBUG", "HPN_GZ_FIND", "HPN_GZ_GPU_FORCE", "HPN_GZ_MEMBERS",
              "HPN_GZ_OVERLAP", "HPN_GZ_OVERSUB", "HPN_GZ_STRETCH", "HPN_GZ_WINDOWS", "HPN_K1L_BIG", "HPN_K1_VARIANT", "HPN_K1_WG_PER_CU", "HPN_NO_BGZF", "HPN_NO_MGZ",
              "HPN_NO_PGZ", "HPN_PGZ_CHUNK", "HPN_PGZ_FORCE", "HPN_READER_STATS", "HPN_SWEEP_DIAG", "HPN_TEXT_CHUNK", "HPN_TEXT_SLICE",
              "HPN_TRIM_WG_PER_CU"}

import subprocess as _sp

_real_popen_init = _sp.Popen.__init__


def _popen_init(self, args, *a, **kw):
    env = kw.get("env")
    if env is not None and not set(env).isdisjoint(TEST_KNOBS):
        if isinstance(args, (list, tuple)) and args and isinstance(args[0], str) and os.path.dirname(os.path.abspath(args[0])) == BIN:
            args = [os.path.join(HOOKS_BIN, os.path.basename(args[0]))] + list(args[1:])
        elif "HPN_LIB" not in env:
            kw["env"] = {**env, "HPN_LIB": HOOKS_LIB}
    _real_popen_init(self, args, *a, **kw)


_sp.Popen.__init__ = _popen_init      # (subprocess.run / check_call / check_output all start their child through Popen)


def in_hooks_build(request, env):
    """For a test whose BODY needs a test switch inside this process's library (the binding loads one library per process):
    the outer call re-runs the very same test item in a child pytest with the hooks library and the switch set, and returns True
    (nothing more to do); the inner call returns False and the body runs."""
    if os.environ.get("HPN_TEST_INNER"):
        return False
    p = _sp.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", request.node.nodeid], cwd=ROOT,
                env={**os.environ, **env, "HPN_LIB": HOOKS_LIB, "HPN_TEST_INNER": "1"}, stdout=_sp.PIPE, stderr=_sp.STDOUT, timeout=1800)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    return True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        m = json.load(f)
    return {c["name"]: c for c in m["cases"]}


def golden_path(*parts):
    return os.path.join(GOLDEN, *parts)


def expected(case, name="stdout"):
    """Bytes the reference tool produced; reports over 1 MiB are stored gzip-compressed."""
    path = os.path.join(GOLDEN, "expected", case, name)
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        import gzip
        with gzip.open(path + ".gz", "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def rccl_stub():
    """Path of the test-only librccl stand-in (tests/stub/rccl_stub.cpp), built on first use: HPN_RCCL_LIB=<path> makes
    csrc/hpn_comm.hip bind it instead of RCCL, so the grouped collective runs with n > 1 'ranks' on a one-GPU box."""
    import subprocess
    src = os.path.join(ROOT, "tests", "stub", "rccl_stub.cpp")
    out = os.path.join(ROOT, "tests", "stub", "librccl_stub.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-shared", "-fPIC", "-O1", "-std=c++17", "-I/opt/rocm/include", src, "-o", out,
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return out
