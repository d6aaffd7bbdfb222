"""GPU probe: pf_conv1x1_split3 on FRESHLY uploaded operands (cold L2 / memory-side cache) at a few small shapes, default grid and 3 / 1 tile slots, against
float64 -- the script that showed the register-prefetch form of the kernel to be wrong on its first launch (whole token rows off by O(1)) and exact on warm
re-runs (profiles/r6_conv1x1_split3.md).  PF_LIB_PATH=.../libpf_c1dbg.so PF_C1_DBG=64 forces every wait to vmcnt(0).  usage: python tools/c1_debug.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PF_CONV1X1_SPLIT3"] = "2"
from patchfusion_amd import packing as pk
from patchfusion_amd.hip_ops import ops
g = torch.Generator().manual_seed(3)
for (M, K, N) in ((4160, 64, 80), (3108, 256, 128), (700, 64, 128), (130, 64, 128)):
    w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
    pw = pk.pack_conv(w, torch.randn(N, generator=g), dtype=torch.float32).to("cuda")
    x = torch.randn(1, 1, M, K, generator=g).to("cuda")
    ref = (x.view(M, K).double() @ w.view(N, K).double().t().cuda() + pw.bias[:N].double())
    for slots in (None, None, "3", "1"):
        if slots: os.environ["PF_C1_SLOTS"] = slots
        else: os.environ.pop("PF_C1_SLOTS", None)
        y = torch.full((1, 1, M, N), -7.0, device="cuda")
        ops.conv(x, pw, y, _direct=False)
        torch.cuda.synchronize()
        d = (y.view(M, N).double() - ref).abs()
        bad = (d > 1e-4).any(dim=1).nonzero().flatten()
        print(M, K, N, "slots", slots, "max err", float(d.max()), "bad rows", bad.numel(), bad[:6].tolist(), bad[-3:].tolist() if bad.numel() else "")
