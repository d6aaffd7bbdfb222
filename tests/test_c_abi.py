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


def test_header_is_plain_c_and_a_c_host_links_against_the_library(tmp_path):
    """the boundary is a C ABI: include/pf_hip.h compiles as C11 with gcc -pedantic (no C++, no torch types) and a C translation unit
    that takes the address of EVERY declared entry point links against libpf_hip.so (what a cgo / JNI / ctypes-free host would do)"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(L.LIB_PATH):
        pytest.skip("gcc or libpf_hip.so not available")
    names = _declared()
    src = tmp_path / "host.c"
    src.write_text('#include "pf_hip.h"\n#include <stdio.h>\n'
                   "typedef void (*fn)(void);\n"
                   "int main(void) {\n  fn table[] = {\n" + "".join(f"    (fn){n},\n" for n in names) +
                   "  };\n  unsigned i, n = 0;\n  for (i = 0; i < sizeof table / sizeof table[0]; ++i) n += table[i] != 0;\n"
                   '  printf("%u entry points, ABI version %d\\n", n, pf_version());\n  return n == sizeof table / sizeof table[0] ? 0 : 1;\n}\n')
    exe = tmp_path / "host"
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        L.LIB_PATH, f"-Wl,-rpath,{os.path.dirname(L.LIB_PATH)}", "-Wl,--unresolved-symbols=ignore-in-shared-libs"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    # ... and runs (no GPU needed: no kernel is launched): every entry point resolved, the version call answers
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and f"{len(names)} entry points, ABI version" in r.stdout, (r.stdout, r.stderr[-1000:])
