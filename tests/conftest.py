import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
