"""pytest -m gpu: every hand-written HIP kernel (through the C ABI) against the plain-PyTorch fp32
reference of the same op, in both compute modes.  Tolerances (relative to max|ref|, floor 1):
fp32 mode 2e-4 for MFMA ops / 1e-5 for pointwise ops; bf16 mode 2e-2 (bf16 storage rounding 2^-8)."""
import pytest
import torch

from tests import op_checks

pytestmark = pytest.mark.gpu

CASES = [(n, d) for n in op_checks.CHECKS for d in ("fp32", "bf16") if not (d == "bf16" and n in op_checks.F32_ONLY)]


@pytest.mark.parametrize("name,dt", CASES)
def test_op(name, dt):
    assert torch.cuda.is_available()
    err, tol, info = op_checks.CHECKS[name](op_checks.DTYPES[dt])
    assert err <= tol, f"{name}[{dt}] {info}: err {err:.3e} > tol {tol:.1e}"


def test_native_library_is_loaded_and_used():
    import patchfusion_amd._lib as L
    lib = L.load()
    assert lib.pf_version() >= 1
    maps = open("/proc/self/maps").read()
    assert "libpf_hip.so" in maps


def test_library_first_import_order_still_launches():
    """A fresh process that touches the kernel library BEFORE anything else of the package (what `build()` followed by `smoke()` does):
    libpf_hip.so must end up on the HIP runtime PyTorch-ROCm ships, otherwise every launch on a torch stream fails
    (patchfusion_amd/_lib.py imports torch ahead of the dlopen for this reason)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import patchfusion_amd._lib as L; L.load(); import torch; from patchfusion_amd.hip_ops import ops; "
            "img = torch.rand(1, 3, 28, 42, device='cuda'); col = torch.zeros(6, 592, device='cuda'); ops.patch_im2col(img, col); "
            "torch.cuda.synchronize(); assert float(col.abs().sum()) > 0; print('launched')")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "launched" in r.stdout, r.stderr[-1500:]
