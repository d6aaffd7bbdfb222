"""GPU probe: pf_roi_align on the shapes of one ViT-L tile batch (coarse map 1 x H x W x C -> 8 ROIs of a 1/16 region at full size): time and
algorithmic GB/s (ROI region of the source read once + output written once) against 8 TB/s.  usage: python tools/roi_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd.hip_ops import ops        # noqa: E402

print("| roi_align (f32) | us | algorithmic GB/s | of 8 TB/s |")
print("|---|---|---|---|")
for (H, W, C) in ((224, 296, 256), (112, 148, 256), (392, 518, 32), (56, 74, 256)):
    feat = torch.randn(1, H, W, C, device="cuda")
    rois = torch.tensor([[0, (i % 4) * 518 / 4.0, (i // 4) * 392 / 4.0, (i % 4 + 1) * 518 / 4.0, (i // 4 + 1) * 392 / 4.0] for i in range(8)], device="cuda")
    y = torch.empty(8, H, W, C, device="cuda")
    scale = H / 392.0
    for _ in range(3):
        ops.roi_align(feat, rois, y, scale)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.roi_align(feat, rois, y, scale)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    byts = y.numel() * 4 + feat.numel() * 4 * 8 / 16
    print(f"| (1, {H}, {W}, {C}) -> (8, {H}, {W}, {C}) | {us:.1f} | {byts / us / 1e3:.0f} | {byts / us / 1e3 / 8000:.3f} |")
