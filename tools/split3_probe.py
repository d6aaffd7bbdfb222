"""GPU probe: the split-precision ("f32x3") linear layer against the f32 MFMA GEMM on the ViT-L linear shapes (time + error vs float64).
usage: python tools/split3_probe.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk       # noqa: E402
from patchfusion_amd.hip_ops import ops        # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [8 * 1037, 4 * 1037]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    print("| M | K -> N | f32 kernel ms (TF/s) | split ms (eff. TF/s; bf16 TF/s executed) | tile | speed-up | err f32 | err split |")
    print("|---|---|---|---|---|---|---|---|")
    for M in Ms:
        for K, N, act in [(1024, 3072, None), (1024, 1024, None), (1024, 4096, "gelu"), (4096, 1024, None)]:
            w = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            x = torch.randn(M, K, generator=g).to(dev)
            pw3 = pk.pack_conv_split3(w, b).to(dev)
            pw = pk.pack_conv(w.view(N, K, 1, 1), b, dtype=torch.float32).to(dev)
            x3 = torch.empty(3, M, K, dtype=torch.bfloat16, device=dev)
            ops.split3(x, x3)
            y = torch.zeros(M, N, device=dev)
            yf = torch.zeros(1, 1, M, N, device=dev)
            ref = x[:512].double() @ w.double().t().to(dev) + b.double().to(dev)
            if act == "gelu":
                ref = torch.nn.functional.gelu(ref)
            den = max(1.0, float(ref.abs().max()))
            tf = ops.conv(x.view(1, 1, M, K), pw, yf, act=act, _timed=20)
            ops.conv(x.view(1, 1, M, K), pw, yf, act=act)
            ef = float((yf.view(M, N)[:512].double() - ref).abs().max()) / den
            best = None
            for tile in ("64", "128"):
                os.environ["PF_S3_TILE_NOW"] = tile
                t3 = ops.conv_split3(x3, pw3, y, act=act, _timed=20)
                ops.conv_split3(x3, pw3, y, act=act)
                e3 = float((y[:512].double() - ref).abs().max()) / den
                if best is None or t3 < best[0]:
                    best = (t3, tile, e3)
            os.environ.pop("PF_S3_TILE_NOW", None)
            t3, tile, e3 = best
            fl = 2.0 * M * K * N
            print(f"| {M} | {K} -> {N} | {tf:.3f} ({fl / tf / 1e9:.1f}) | {t3:.3f} ({fl / t3 / 1e9:.1f}; {6 * fl / t3 / 1e9:.0f}) | {tile} | {tf / t3:.2f}x | {ef:.2e} | {e3:.2e} |")
        # the split itself (stand-alone form)
        x = torch.randn(M, 1024, generator=g).to(dev)
        x3 = torch.empty(3, M, 1024, dtype=torch.bfloat16, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.split3(x, x3)
        e0.record()
        for _ in range(20):
            ops.split3(x, x3)
        e1.record()
        torch.cuda.synchronize()
        print(f"\nsplit3 of [{M}, 1024]: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us\n")


if __name__ == "__main__":
    main()
