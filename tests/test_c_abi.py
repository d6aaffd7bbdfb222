"""CPU-side checks of the drop-in boundary: libpf_hip.so loads and exports every symbol that
include/pf_hip.h declares (no compute calls without a GPU), and argument validation returns
status codes instead of launching."""
import ctypes
import os
import re

import pytest

import patchfusion_amd._lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "pf_hip.h")).read()
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("libpf_hip.so not built (run python __graft_entry__.py)")
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pf_hip.h but not exported"
    # and the ctypes table binds exactly the declared compute entry points
    assert set(L.SIGNATURES) == set(names) - set(L.NON_STATUS)


def test_conv_argument_validation_without_gpu():
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("libpf_hip.so not built")
    lib = L.load()
    assert lib.pf_version() >= 1
    p = L.ConvParams()
    assert lib.pf_conv(ctypes.byref(p), None) == 1          # PF_ERR_ARG: null tensors
    assert b"null" in lib.pf_last_error()
    assert lib.pf_layernorm(None, 8, None, 8, None, None, 1e-6, 1, 1, 0, 1, 8, 0, None) == 1
