"""Run the ViT attention entry point alone (ViT-L shape: B=8, S=1037, 16 heads, head_dim 64) so that rocprofv3 --pmc can be
pointed at vit_attention32_kernel / vit_attention_kernel<float>.   usage: python tools/attn_probe.py <bf16|fp32> [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd.hip_ops import ops  # noqa: E402

if sys.argv[1] == "split3":          # float32-grade attention on the bf16 matrix cores: q / k / v and the output as three bf16 planes
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    B, S, H = 8, 1037, 16
    qkv = torch.randn(B * S, 3 * H * 64, device="cuda") * 0.5
    q3 = torch.empty(3, B * S, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
    ops.split3(qkv, q3)
    o3 = torch.empty(3, B * S, H * 64, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):
        ops.vit_attention(q3, o3, B, S, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.vit_attention(q3, o3, B, S, H)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 4.0 * B * H * S * S * 64
    print(f"vit_attention split3 (planes in, planes out): {us:.1f} us per call = {fl / us / 1e6:.1f} TFLOP/s useful")
    sys.exit(0)
dt = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, S, H = 8, 1037, 16
qkv = (torch.randn(B * S, 3 * H * 64, device="cuda") * 0.5).to(dt)
out = torch.empty(B * S, H * 64, device="cuda", dtype=dt)
for _ in range(2):
    ops.vit_attention(qkv, out, B, S, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.vit_attention(qkv, out, B, S, H)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
fl = 4.0 * B * H * S * S * 64
print(f"vit_attention {sys.argv[1]} (qkv_split + attention): {us:.1f} us per call = {fl / us / 1e6:.1f} TFLOP/s")
