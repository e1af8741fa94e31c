"""CPU: the C-ABI library loads and exports every symbol include/hpngs.h declares."""
import ctypes as C
import os
import re

import pytest

from highperformancengs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "hpngs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hpn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 30
    bound = {s[0] for s in _lib.SYMBOLS}
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hpngs.h but not exported by libhpngs.so"
        assert n in bound, f"{n} has no ctypes signature in _lib.SYMBOLS"
    assert bound <= set(names)


def test_layout_constants_match_header():
    text = open(os.path.join(ROOT, "include", "hpngs.h")).read()
    assert f"#define HPN_ABI_VERSION {_lib.lib().hpn_abi_version()}" in text
    assert _lib.TALLY_WORDS == 516 + 128 * 512 + 5 * 512
    assert C.sizeof(_lib.Tally) == 8 * 512 + 3 * 8 + 2 * 8
    assert C.sizeof(_lib.Run) == 12 and C.sizeof(_lib.BamBatch) == 9 * 8


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import highperformancengs_amd as hp
    with pytest.raises(hp.HpnError) as e:
        hp.Context(0)
    assert e.value.status == _lib.E_NODEVICE


def test_product_does_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under the package may name it
    pkg = os.path.join(ROOT, "highperformancengs_amd")
    for dp, _, fs in os.walk(pkg):
        if os.sep + "build" in dp:
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                assert "hpn_oracle" not in t and "liborc" not in t and "import orc" not in t, os.path.join(dp, f)
