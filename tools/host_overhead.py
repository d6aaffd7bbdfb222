"""CPU-only probe of the HOST side of one image pass: every C entry point is replaced by a no-op and the pass runs on CPU tensors, so what is
timed is the Python of model.py / engine.py / hip_ops.py (argument checks, ctypes structs, allocations) -- the floor the GPU queue must be fed at.
usage: python tools/host_overhead.py [vits|vitl]    (round 3: vits 32 ms per pass = 22 us per library call; the vits pass takes 44 ms on the GPU)"""
import sys, time, types
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchfusion_amd._lib as L
lib = L.load()
import patchfusion_amd.hip_ops as H
# mock: every C entry point returns 0 immediately; pointer / stream helpers accept CPU tensors
class FakeLib:
    def __getattr__(self, n):
        if n == "pf_conv_winograd_fused_supported":
            return lambda *a: 1
        return lambda *a: 0
H._L = FakeLib()
H._p = lambda t: None if t is None else t.data_ptr()
H._stream = lambda: None
_ws = {}
def _fake_ws(device, nV, nM):
    k = 0
    V, M = _ws.get(k, (None, None))
    if V is None or V.numel() < nV: V = torch.empty(nV)
    if M is None or M.numel() < nM: M = torch.empty(nM)
    _ws[k] = (V, M)
    return V, M
H._workspace = _fake_ws
from patchfusion_amd.config import make_config
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
enc = sys.argv[1] if len(sys.argv) > 1 else "vits"
cfg = make_config(enc, (392, 518), (2160, 3840), (4, 4))
m = PatchFusion(cfg, compute_dtype="fp32", ops=H.ops).eval()
m.load_state_dict(synthetic_state_dict(patchfusion_spec(cfg), 0), strict=True)
img = torch.rand(1, 3, 2160, 3840)
lr = m.resizer(img)
calls = [0]
orig = FakeLib.__getattr__
def counting(self, n):
    f = orig(self, n)
    def g(*a):
        calls[0] += 1
        return f(*a)
    return g
FakeLib.__getattr__ = counting
with torch.no_grad():
    for it in range(3):
        calls[0] = 0
        t0 = time.perf_counter()
        m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=4 if enc == "vits" else 8)
        dt = time.perf_counter() - t0
        print(f"{enc}: host time of one pass with no-op kernels: {dt * 1e3:.1f} ms, {calls[0]} library calls -> {dt / calls[0] * 1e6:.1f} us per call")
