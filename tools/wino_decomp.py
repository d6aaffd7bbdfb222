"""GPU probe: where the time of a three-step Winograd layer goes -- whole layer (input transform -> batched split GEMM -> output transform, as the pass runs it, windows
included) against its GEMM launches alone, for the layers of the 4K pass with 256 output channels (0.20-0.31 of 2500/6 in profiles/r5_op_roofline_fp32.md) next to
the two big ones.  Algorithmic HBM bytes of the transforms beside their time.   usage: python tools/wino_decomp.py   (profiles/r6_wino_n256_decomp.md)"""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd import hip_ops, packing as pk
from patchfusion_amd.hip_ops import ops
hip_ops.refresh_env()
dev = "cuda"
print("| layer | tiles (window x n) | layer ms | GEMM ms (all windows) | transforms ms | GEMM useful TF/s (of 416.7) | transform bytes GB (in+V / M+out) | transforms at GB/s | layer of 416.7 |")
print("|---|---|---|---|---|---|---|---|---|")
for (cin, cout, B, H, W) in ((544, 544, 8, 392, 518), (768, 768, 8, 224, 296), (768, 256, 8, 224, 296), (512, 256, 8, 224, 296), (256, 256, 8, 224, 296), (256, 256, 8, 196, 259),
                             (768, 256, 8, 112, 148), (512, 256, 8, 112, 148), (256, 256, 8, 112, 148), (256, 256, 8, 98, 129), (512, 256, 8, 56, 74)):
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.float32).to(dev)
    x = torch.randn(B, H, W, cin, device=dev)
    y = torch.empty(B, H, W, cout, device=dev)
    ms_layer = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
    T, ns = hip_ops.wino3_window(B, H, W, pw)[:2]
    Tall = B * -(-H // 4) * -(-W // 4)
    T = min(T, Tall)
    V3 = torch.randn(3, 36, cin // 32, T, 32, device=dev).to(torch.bfloat16)
    Mw = torch.empty(36 * T * cout, device=dev)
    ms = ops.gemm_planes_split3_timed(V3, pw.wino_u3, Mw.view(36, T, cout), T, cin, cout, 5 * ns)
    g_all = ms * Tall / T
    fl = 36 * 2.0 * Tall * cin * cout
    b_in = B * H * W * cin * 4 + 36 * Tall * cin * 6
    b_out = 36 * Tall * cout * 4 + B * H * W * cout * 4
    tr = ms_layer - g_all
    print(f"| {cin}->{cout} @ {B}x{H}x{W} | {T} x {ns} | {ms_layer:.3f} | {g_all:.3f} | {tr:.3f} | {fl / g_all / 1e9:.1f} ({fl / g_all / 1e9 / 416.7:.3f}) | {b_in / 1e9:.2f} / {b_out / 1e9:.2f} | "
          f"{(b_in + b_out) / tr / 1e6:.0f} | {fl / ms_layer / 1e9 / 416.7:.3f} |")
    del V3, Mw, x, y, pw
    torch.cuda.empty_cache()
