"""GPU probe: bilinear resize / resize_concat kernels at the image pass's shapes, version 2 (source-aligned) vs the output-walking kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd.hip_ops import ops        # noqa: E402


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    cases = [("concat 256+256 (224,296)->(392,518)", [(224, 296, 256), (224, 296, 256)], (392, 518)),
             ("concat 256x3 (112,148)->(224,296)", [(112, 148, 256)] * 3, (224, 296)),
             ("concat 256x3 (56,74)->(112,148)", [(56, 74, 256)] * 3, (112, 148)),
             ("resize 128 (224,296)->(392,518)", [(224, 296, 128)], (392, 518)),
             ("resize 256 (112,148)->(224,296)", [(112, 148, 256)], (224, 296)),
             ("resize 128 (112,148)->(224,296)", [(112, 148, 128)], (224, 296))]
    for name, srcs, (oh, ow) in cases:
        xs = [torch.randn(8, h, w, c, device="cuda").to(dt) for h, w, c in srcs]
        ct = sum(c for _, _, c in srcs)
        y = torch.empty(8, oh, ow, ct, device="cuda", dtype=dt)
        by = (sum(x.numel() for x in xs) + y.numel()) * es
        row = []
        for v2 in ("1", "0"):
            os.environ["PF_RESIZE_V2"] = v2
            us = timed((lambda: ops.resize_concat(xs, y)) if len(xs) > 1 else (lambda: ops.resize(xs[0], y)))
            row.append(f"v{'2' if v2 == '1' else '1'} {us:7.1f} us {by / us / 1e3:6.0f} GB/s ({by / us / 1e3 / 80:.0f} %)")
        print(f"{name:40s} " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
