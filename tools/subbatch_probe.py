"""GPU probe (round 5): the dominant batched GEMM launch (36 x [T x 544].[544 x 544], chunk-major planes) at the token counts of 8 / 4 / 2 / 1 tiles of
392 x 518, and the whole three-step layer through the workspace cap that produces those launches -- what a sub-batched layer pays in the GEMM and
gains in the transforms.   usage: python tools/subbatch_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import hip_ops, packing as pk   # noqa: E402
from patchfusion_amd.hip_ops import ops             # noqa: E402

DEV = "cuda"
C = 544
g = torch.Generator().manual_seed(0)
w = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
pw = pk.pack_conv(w, torch.zeros(C), dtype=torch.float32).to(DEV)
x = torch.randn(8, 392, 518, C, device=DEV)
y = torch.empty(8, 392, 518, C, device=DEV)
print("| tiles per launch | T | GEMM ms / launch | useful TF/s | of 2500/6 | x launches for 8 tiles = ms | whole layer over 8 tiles at that sub-batch ms | transforms ms |")
print("|---|---|---|---|---|---|---|---|")
for b, cap in ((8, "100"), (4, "9.4"), (2, "4.7"), (1, "2.4")):
    T = b * 98 * 130
    V3 = torch.randn(3, 36, C // 32, T, 32, device=DEV).to(torch.bfloat16)
    Mw = torch.empty(36, T, C, device=DEV)
    ops.gemm_planes_split3(V3, pw.wino_u3, Mw, T, C, C, 3)
    ms = ops.gemm_planes_split3(V3, pw.wino_u3, Mw, T, C, C, 40 // b)
    del V3, Mw
    os.environ["PF_WS_CAP_GB"] = cap
    hip_ops.refresh_env()
    hip_ops.release_workspaces()
    torch.cuda.empty_cache()
    assert hip_ops.wino3_window(8, 392, 518, pw)[1] == 8 // b, hip_ops.wino3_window(8, 392, 518, pw)
    ops.conv(x, pw, y, pad=1, act="relu", _timed=2)
    layer = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
    fl = 36 * 2.0 * T * C * C
    print(f"| {b} | {T} | {ms:.3f} | {fl / ms / 1e9:.1f} | {fl / ms / 1e9 / (2500 / 6):.3f} | {8 // b * ms:.3f} | {layer:.3f} | {layer - 8 // b * ms:.3f} |", flush=True)
