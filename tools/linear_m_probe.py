"""Does the f32 ViT-L linear rate depend on the token count per launch?  M = 8296 (8 tiles, what process_num=8 gives), 16592 (all 16 tiles
of the image in one encoder batch), 4148.  usage: python tools/linear_m_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk  # noqa: E402
from patchfusion_amd.hip_ops import ops  # noqa: E402

dev = torch.device("cuda", 0)
for name, cin, cout in (("qkv", 1024, 3072), ("proj", 1024, 1024), ("fc1", 1024, 4096), ("fc2", 4096, 1024)):
    pw = pk.pack_conv(torch.randn(cout, cin) / cin ** 0.5, torch.zeros(cout), dtype=torch.float32).to(dev)
    out = []
    for M in (4148, 8296, 16592, 33184):
        x = torch.randn(M, cin, device=dev)
        y = torch.empty(M, cout, device=dev)
        best = {}
        for cfg in ("", "3", "6", "7"):
            if cfg:
                os.environ["PF_IGEMM_CFG"] = cfg
            else:
                os.environ.pop("PF_IGEMM_CFG", None)
            ms = ops.conv(x, pw, y, _timed=5)
            best[cfg or "auto"] = 2.0 * M * cin * cout / ms / 1e9
        os.environ.pop("PF_IGEMM_CFG", None)
        out.append(f"M={M}: " + " ".join(f"{k}:{v:.0f}" for k, v in best.items()))
    print(f"{name} {cin}->{cout} TF/s  " + " | ".join(out), flush=True)
