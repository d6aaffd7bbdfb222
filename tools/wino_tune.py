"""Kernel-tuning aid for the float32 Winograd layers (one process, GPU): (1) the batched transform-domain GEMM of the main layer
shapes under the tile / split overrides of the generic kernel (PF_IGEMM_CFG, PF_IGEMM_NOSPLIT are read per call); (2) whole layers,
Winograd F(4x4,3x3) vs the direct kernel, around the PF_WINOGRAD_MIN_PIXELS threshold.   usage: python tools/wino_tune.py > log"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PF_WINOGRAD"] = "4"
from patchfusion_amd import packing as pk  # noqa: E402
from patchfusion_amd.hip_ops import ops  # noqa: E402

dev = torch.device("cuda", 0)
CFGS = ({}, {"PF_IGEMM_NOSPLIT": "1"}, {"PF_IGEMM_CFG": "1"}, {"PF_IGEMM_CFG": "2"}, {"PF_IGEMM_CFG": "3"}, {"PF_IGEMM_CFG": "6"}, {"PF_IGEMM_CFG": "7"})


def gemm(T, cin, cout, tag):
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    pw = pk.pack_conv(w, None, dtype=torch.float32).to(dev)
    V = torch.randn(36 * T * cin, device=dev)
    M = torch.empty(36 * T * cout, device=dev)
    fl = 36 * 2.0 * T * cin * cout
    res = []
    for env in CFGS:
        for k in ("PF_IGEMM_NOSPLIT", "PF_IGEMM_CFG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            ms = ops.gemm_planes_timed(V, pw.wino_u, M, 36, T, cin, cout, 3)
            res.append(f"{'+'.join(f'{k[9:]}={v}' for k, v in env.items()) or 'auto'}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF/s")
        except Exception as e:
            res.append(f"{env}: {type(e).__name__}")
    for k in ("PF_IGEMM_NOSPLIT", "PF_IGEMM_CFG"):
        os.environ.pop(k, None)
    print(f"gemm {tag} T={T} {cin}->{cout}: " + " | ".join(res), flush=True)


def layer(B, H, W, cin, cout):
    os.environ["PF_WINOGRAD_MIN_PIXELS"] = "0"
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.float32).to(dev)
    x = torch.randn(B, H, W, cin, device=dev)
    y = torch.empty(B, H, W, cout, device=dev)
    tw = ops.conv(x, pw, y, pad=1, act="relu", _timed=3) if pw.wino_u is not None else float("nan")
    td = ops.conv(x, pw, y, pad=1, act="relu", _timed=3, _direct=True)
    print(f"layer ({B},{H},{W}) {cin}->{cout}: pixels {B * H * W}  winograd {tw:.3f} ms  direct {td:.3f} ms  ratio {td / tw:.2f}", flush=True)


CFGS = ({}, {"PF_IGEMM_CFG": "3"}, {"PF_IGEMM_CFG": "4"}, {"PF_IGEMM_CFG": "5"}, {"PF_IGEMM_CFG": "6"})
gemm(8 * 98 * 130, 544, 32, "544->32 @392x518")
gemm(8 * 98 * 130, 128, 32, "128->32 @392x518")
gemm(8 * 98 * 130, 544, 544, "fusion up-conv (auto = the batch rule)")
for shp in ((8, 392, 518, 544, 32), (8, 392, 518, 128, 32), (8, 56, 74, 768, 256), (8, 49, 64, 256, 256), (8, 24, 32, 256, 256), (8, 14, 19, 1024, 256),
            (8, 14, 19, 256, 256), (8, 12, 16, 256, 256), (1, 112, 148, 256, 256), (1, 56, 74, 256, 256), (1, 56, 74, 512, 256), (1, 28, 37, 1024, 256),
            (1, 28, 37, 256, 256), (1, 14, 19, 1024, 256), (1, 14, 19, 256, 256), (1, 392, 518, 128, 32)):
    layer(*shp)
