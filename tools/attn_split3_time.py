"""GPU probe: the split-precision ViT attention launch alone (csrc/vit.hip vit_attention_split3_kernel), B = 8 and B = 1 x 16 heads x 1037 tokens, random
bf16 planes in, chunk-major planes out; HIP-event average of 50 launches.  A/B against another build: PF_LIB_PATH=<other libpf_hip.so>.
usage: python tools/attn_split3_time.py   (profiles/r4_attention_swizzle.md)"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd.hip_ops import ops
g = torch.Generator().manual_seed(0)
for B in (8, 1):
    S, H = 1037, 16
    q3 = torch.randn(3, B * S, 3 * H * 64, generator=g).to(torch.bfloat16).cuda()
    out = torch.empty(3, H * 64 // 32, B * S, 32, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.vit_attention(q3, out, B, S, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.vit_attention(q3, out, B, S, H)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"vit_attention_split3 B{B}: {us:.1f} us = {4.0*B*H*S*S*64/us/1e6:.1f} TF/s useful")
