"""Achieved HBM bandwidth of the input / output side kernels (io.hip) at the BASELINE geometry
(1080p photo -> 2160x3840 image_hr; stitched 1568x2072 depth; 4K ground truth).  Prints one JSON object;
`GB/s` = algorithmic bytes (what must cross HBM once) / measured time, `frac` = / 8 TB/s."""
import json
import sys

import torch

sys.path.insert(0, ".")
from patchfusion_amd import postprocess as post
from patchfusion_amd.hip_ops import ops
from patchfusion_amd.preprocess import ImagePreprocessor


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    res = {}
    img = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, generator=g).to(dev)
    pre = ImagePreprocessor((2160, 3840), (392, 518))
    hr = torch.empty((3, 2160, 3840), dtype=torch.float32, device=dev)
    t = timed(lambda: ops.u8_bicubic_to_f32(img, hr))
    res["u8_bicubic_1080p_to_4k"] = (img.numel() + hr.numel() * 4, t)
    img4k = torch.randint(0, 256, (2160, 3840, 3), dtype=torch.uint8, generator=g).to(dev)
    t = timed(lambda: ops.u8_bicubic_to_f32(img4k, hr))
    res["u8_to_f32_4k_same_size"] = (img4k.numel() + hr.numel() * 4, t)
    t = timed(lambda: pre(img))
    res["preprocess_total(hr+lr)"] = (img.numel() + hr.numel() * 4 * 2 + 3 * 392 * 518 * 4, t)
    d = (torch.rand(1568, 2072, generator=g) ** 2 * 40).to(dev)
    t = timed(lambda: ops.percentiles(d, 2, 95, invalid_val=-99))
    res["percentiles_2_95 (3 passes)"] = (d.numel() * 4, t)
    t = timed(lambda: post.colorize(d, cmap="magma_r"))
    res["colorize incl. percentiles"] = (d.numel() * 8, t)
    t = timed(lambda: post.depth_to_uint16(d))
    res["depth_to_uint16"] = (d.numel() * 6, t)
    gt = (torch.rand(2160, 3840, generator=g) * 40 + 0.5).to(dev)
    ed = (torch.rand(2160, 3840, generator=g) < 0.05).float().to(dev)
    out = torch.empty(13, dtype=torch.float64, device=dev)
    t = timed(lambda: ops.depth_metrics(gt, d, ed, 1e-3, 80, (0, 2160, 0, 3840), out))
    res["depth_metrics 4K gt + edges"] = (gt.numel() * 8 + d.numel() * 4, t)
    print(json.dumps({k: {"algorithmic_MB": round(b / 1e6, 1), "us": round(t * 1e6, 1), "GB/s": round(b / t / 1e9, 1), "frac_of_8TBs": round(b / t / 8e12, 3)}
                      for k, (b, t) in res.items()}, indent=1))


if __name__ == "__main__":
    main()
