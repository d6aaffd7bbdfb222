"""GPU probe: how much of a kernel's TFLOP/s is set by the data-dependent power budget (DVFS) rather than by its instruction stream.
The same launches on RANDOM operands (what bench.py uses) and on ZERO operands (identical instruction stream, far fewer bit toggles in the
matrix pipes -> the chip holds a higher clock).  The ratio is the head-room a roofline fraction quoted against the NOMINAL peak (2.4 GHz)
can never reach on real data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk       # noqa: E402
from patchfusion_amd.hip_ops import ops        # noqa: E402


def main():
    dev = "cuda"
    B, H, W, C = 8, 392, 518, 544
    rows = []
    for kind in ("random", "zeros"):
        mk = (lambda *s: torch.randn(*s)) if kind == "random" else (lambda *s: torch.zeros(*s))
        # bf16 halo conv (the bf16 mode's dominant launch)
        w = mk(C, C, 3, 3) / (9 * C) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(C), dtype=torch.bfloat16).to(dev)
        x = mk(B, H, W, C).to(torch.bfloat16).to(dev)
        y = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dev)
        ms = ops.conv(x, pw, y, pad=1, act="relu", _timed=10)
        rows.append((kind, "bf16 3x3 halo conv 544->544 @8x392x518", ms, 2.0 * B * H * W * 9 * C * C / ms / 1e9, 2500.0))
        del x, y
        # f32 fused Winograd (the f32 mode's dominant launch)
        pwf = pk.pack_conv(w, torch.zeros(C), dtype=torch.float32).to(dev)
        x = mk(B, H, W, C).to(dev)
        y = torch.empty(B, H, W, C, device=dev)
        ms = ops.conv(x, pwf, y, pad=1, act="relu", _timed=5)
        T = B * -(-H // 4) * -(-W // 4)
        rows.append((kind, "f32 fused Winograd 544->544 @8x392x518 (executed flops)", ms, 36 * 2.0 * T * C * C / ms / 1e9, 157.3))
        del x, y
        # split-precision GEMM (executed bf16 flops = 6 x)
        M, K, N = 8296, 1024, 3072
        pw3 = pk.pack_conv_split3(mk(N, K) / K ** 0.5, torch.zeros(N)).to(dev)
        x3 = torch.empty(3, M, K, dtype=torch.bfloat16, device=dev)
        ops.split3(mk(M, K).to(dev), x3)
        yy = torch.empty(M, N, device=dev)
        ms = ops.conv_split3(x3, pw3, yy, _timed=20)
        rows.append((kind, "split GEMM 8296 x 1024 -> 3072 (executed bf16 flops)", ms, 6 * 2.0 * M * K * N / ms / 1e9, 2500.0))
    print("| operands | launch | ms | TFLOP/s | of nominal peak |")
    print("|---|---|---:|---:|---:|")
    for kind, name, ms, tf, peak in rows:
        print(f"| {kind} | {name} | {ms:.3f} | {tf:.1f} | {tf / peak:.3f} |")


if __name__ == "__main__":
    main()
