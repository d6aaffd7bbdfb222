"""GPU probe: the fused metric-bins tail (pf_bins_tail) against the four launches it replaces, at the image pass's size (8 x 392 x 518)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk       # noqa: E402
from patchfusion_amd.hip_ops import ops        # noqa: E402


def timed(fn, iters=10):
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
    dev = "cuda"
    B, H, W, he, we = 8, 392, 518, 224, 296
    for ctot in (168, 160):
        mlp0 = pk.pack_conv(torch.randn(80, ctot, 1, 1) / ctot ** 0.5, torch.randn(80) * 0.1, dtype=torch.float32).to(dev)
        mlp2 = pk.pack_conv(torch.randn(4, 80, 1, 1) / 80 ** 0.5, torch.randn(4) * 0.1, dtype=torch.float32).to(dev)
        tw = pk.bins_tail_weights(mlp0, mlp2, 128).to(dev)
        clb = torch.randn(B, H, W, ctot, device=dev)
        emb = torch.randn(B, he, we, 128, device=dev)
        cen = (torch.rand(B, he, we, 64, device=dev) * 5 + 0.5).sort(-1).values.contiguous()
        d = torch.empty(B, H, W, device=dev)
        t = torch.empty(B, H, W, 80, device=dev)
        pt = torch.empty(B, H, W, 4, device=dev)
        us_f = timed(lambda: ops.bins_tail(clb, emb, tw, cen, d, 0.0212, 50.0))
        parts = [timed(lambda: ops.resize(emb, clb[..., 32:160])), timed(lambda: ops.conv(clb, mlp0, t, act="gelu")),
                 timed(lambda: ops.conv(t, mlp2, pt, act="softplus")), timed(lambda: ops.logbinom_depth(pt, cen, d, 0.0212, 50.0))]
        fl = 2.0 * B * H * W * (ctot * 80 + 80 * 4)
        print(f"ctot {ctot}: fused {us_f:.1f} us ({fl / us_f / 1e6:.1f} TF/s of the two layers' FLOPs)  vs  resize {parts[0]:.1f} + conv {parts[1]:.1f} + conv {parts[2]:.1f} "
              f"+ logbinom {parts[3]:.1f} = {sum(parts):.1f} us  -> {sum(parts) / us_f:.2f}x", flush=True)


if __name__ == "__main__":
    main()
