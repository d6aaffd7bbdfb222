"""GPU probe (round 4): the persistent tile walk of the split-precision GEMM (csrc/gemm_split3.hip gemm_split3_persist_kernel) against the
one-tile-per-block ping-pong kernel -- the batched transform-domain GEMMs of the three-step Winograd layers and the ViT-L block linears.
usage: python tools/persist_probe.py [wino] [vit]      (PF_S3_PERSIST=0 / 1 is set per measurement; times are HIP-event averages)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk       # noqa: E402
from patchfusion_amd.hip_ops import ops        # noqa: E402

DEV = "cuda"


def timed(fn, modes=("0", "1")):
    out = []
    for m in modes:
        os.environ["PF_S3_PERSIST"] = m
        out.append(fn())
    os.environ.pop("PF_S3_PERSIST", None)
    return out


def wino():
    print("| batched transform-domain GEMM (36 planes) | T | one-tile ms | persistent ms | speed-up | useful TF/s | executed bf16 TF/s (incl. padding) | of 2500/6 |")
    print("|---|---|---|---|---|---|---|---|")
    g = torch.Generator(device=DEV).manual_seed(0)
    for (cin, cout, B, H, W) in ((544, 544, 8, 392, 518), (768, 768, 8, 224, 296), (768, 768, 8, 112, 148), (768, 256, 8, 224, 296),
                                 (512, 256, 8, 224, 296), (256, 256, 8, 224, 296), (768, 768, 8, 56, 74), (544, 544, 4, 392, 518)):
        T = B * -(-H // 4) * -(-W // 4)
        V3 = torch.randn(3, 36, T, cin, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
        rows = -(-cout // 16) * 16
        U3 = (torch.randn(3, 36, rows, cin, device=DEV, generator=g) / cin ** 0.5).to(torch.bfloat16)
        Mw = torch.empty(36, T, cout, device=DEV)
        t0, t1 = timed(lambda: ops.gemm_planes_split3(V3, U3, Mw, T, cin, cout, 5))
        fl = 36 * 2.0 * T * cin * cout
        pad = (-(-cout // 128) * 128) / cout
        print(f"| {cin}->{cout} @ {B}x{H}x{W} | {T} | {t0:.3f} | {t1:.3f} | {t0 / t1:.2f}x | {fl / t1 / 1e9:.1f} | {6 * fl * pad / t1 / 1e9:.0f} | {fl / t1 / 1e9 / (2500 / 6):.3f} |")
        del V3, U3, Mw
        torch.cuda.empty_cache()


def vit():
    print("\n| ViT-L linear | M | one-tile ms | persistent ms | speed-up | useful TF/s | of 2500/6 |")
    print("|---|---|---|---|---|---|---|")
    g = torch.Generator().manual_seed(0)
    for M in (8 * 1037, 1037):
        for name, K, N, act, res, scale, split_out in (("qkv", 1024, 3072, None, False, False, True), ("proj", 1024, 1024, None, True, True, False),
                                                       ("fc1", 1024, 4096, "gelu", False, False, True), ("fc2", 4096, 1024, None, True, True, False)):
            w = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            sc = (0.5 + torch.rand(N, generator=g)) if scale else None
            pw3 = pk.pack_conv_split3(w, b, scale=sc).to(DEV)
            x3 = torch.randn(3, M, K, generator=g).to(torch.bfloat16).to(DEV)
            r = torch.randn(M, N, generator=g).to(DEV) if res else None
            y = torch.empty(3, M, N, dtype=torch.bfloat16, device=DEV) if split_out else torch.empty(M, N, device=DEV)
            t0, t1 = timed(lambda: ops.conv_split3(x3, pw3, y, act=act, res=r, _timed=20), ("0", "2"))
            fl = 2.0 * M * K * N
            print(f"| {name} {K}->{N}{' gelu' if act else ''}{' +res*ls' if res else ''}{' planes out' if split_out else ''} | {M} | {t0:.3f} | {t1:.3f} | {t0 / t1:.2f}x | {fl / t1 / 1e9:.1f} | {fl / t1 / 1e9 / (2500 / 6):.3f} |")


if __name__ == "__main__":
    what = sys.argv[1:] or ["wino", "vit"]
    if "wino" in what:
        wino()
    if "vit" in what:
        vit()
