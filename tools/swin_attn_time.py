"""GPU probe: pf_swin_window_attention (f32) on the G2L shapes of one 4K image (configs[2]): time per launch, algorithmic GB/s (qkv read once +
output written once) against 8 TB/s, and useful f32 TF/s (QK^T + PV).  PF_SWIN_MFMA=0 selects the VALU kernel (rounds 1-5); run the script once
per setting (the switch is read once per process).  usage: [PF_SWIN_MFMA=0] python tools/swin_attn_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd.hip_ops import ops        # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "launch":
    # N launches of ONE level (tools/kernel_pmc.sh): python tools/swin_attn_time.py launch N Hp Wp C heads
    n, Hp, Wp, C, heads = (int(v) for v in sys.argv[2:7])
    qkv = torch.randn(Hp * Wp, 3 * C, device="cuda")
    bt = torch.randn(529, heads, device="cuda") * 0.5
    out = torch.empty(Hp * Wp, C, device="cuda")
    for _ in range(n):
        ops.swin_window_attention(qkv, out, bt, 1, Hp, Wp, C, heads, 6)
    torch.cuda.synchronize()
    sys.exit(0)
mode = "VALU kernel (PF_SWIN_MFMA=0)" if os.environ.get("PF_SWIN_MFMA", "1") == "0" else "f32-MFMA kernel (default)"
print(f"| window attention, {mode} | us | algorithmic GB/s | of 8 TB/s | useful TF/s |")
print("|---|---|---|---|---|")
total = 0.0
# (Hp, Wp, C, heads, launches per image): levels of the coarse feature pyramid, padded to windows of 12
for (Hp, Wp, C, heads, n) in ((396, 528, 32, 8, 2), (228, 300, 256, 8, 2), (120, 156, 256, 16, 3), (60, 84, 256, 16, 3), (36, 48, 256, 16, 3), (24, 24, 256, 16, 3)):
    nt = Hp * Wp
    qkv = torch.randn(nt, 3 * C, device="cuda")
    bt = torch.randn(529, heads, device="cuda") * 0.5
    out = torch.empty(nt, C, device="cuda")
    best = 1e9
    for shift in (0, 6):
        for _ in range(3):
            ops.swin_window_attention(qkv, out, bt, 1, Hp, Wp, C, heads, shift)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.swin_window_attention(qkv, out, bt, 1, Hp, Wp, C, heads, shift)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    byts = nt * C * 16
    flops = 4.0 * nt * 144 * C
    total += best * n
    print(f"| tokens {nt} C{C} heads{heads} (x{n}) | {best:.1f} | {byts / best / 1e3:.0f} | {byts / best / 1e3 / 8000:.3f} | {flops / best / 1e6:.1f} |")
print(f"\nsum over one image's launches: {total / 1e3:.3f} ms")
