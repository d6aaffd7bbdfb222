"""Kernel-tuning aid: time the f32 generic implicit-GEMM kernel on the FLOP-carrying shapes for every tile configuration
(PF_IGEMM_CFG: auto = cost model incl. the exact channel split, 1 = 128x128, 2 = 128x96, 3 = 128x64, 6 = 64x64,
7 = 256x256 eight waves) in ONE process and check that every variant is BIT-IDENTICAL to the default (same K order per
output element, so any difference is a bug).  Logs of the round-2 experiments: profiles/r2_f32_tune*.log.
usage: python tools/f32_tune.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk  # noqa: E402
from patchfusion_amd.hip_ops import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [("c544_544", (8, 392, 518), 544, 544, 3), ("up4_1", (8, 224, 296), 768, 768, 3), ("c256_L3", (8, 112, 148), 256, 256, 3),
          ("qkv", (1, 1, 8296), 1024, 3072, 1), ("fc1", (1, 1, 8296), 1024, 4096, 1), ("fc2", (1, 1, 8296), 4096, 1024, 1),
          ("proj", (1, 1, 8296), 1024, 1024, 1)]
res = []
for name, (B, H, W), cin, cout, k in SHAPES:
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, cin, device=dev)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pw = pk.pack_conv(w, torch.randn(cout, generator=g), dtype=torch.float32).to(dev)
    fl = 2.0 * B * H * W * cin * k * k * cout
    ref = None
    for cfg in ("", "1", "3", "6", "7"):
        if cfg:
            os.environ["PF_IGEMM_CFG"] = cfg
        else:
            os.environ.pop("PF_IGEMM_CFG", None)
        y = torch.empty(B, H, W, cout, device=dev)
        ms = ops.conv(x, pw, y, pad=k // 2, act="relu", _timed=3)
        torch.cuda.synchronize()
        same = True if ref is None else bool(torch.equal(y, ref))
        if ref is None:
            ref = y.clone()
        r = dict(shape=name, cfg=cfg or "auto", ms=ms, tflops=fl / ms / 1e9, frac=fl / ms / 1e9 / 157.3, bit_identical=same)
        res.append(r)
        print(f"{name:10s} cfg {cfg or 'auto':4s}: {ms:8.3f} ms {r['tflops']:7.1f} TF/s ({100 * r['frac']:.1f} %)  {'==' if same else 'DIFFERENT'}", flush=True)
        del y
    del x, ref
os.environ.pop("PF_IGEMM_CFG", None)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=0)
