"""TEST INFRASTRUCTURE: a deterministic stand-in for the EXTERNAL relative-depth core of a type-'ZoeDepth' branch.

The real core (MiDaS DPT_BEiT_L_384) is an un-vendored torch.hub repository (external/zoedepth/models/base_models/midas.py:340)
and cannot be restated offline -- PARITY UNPINNED.  Everything AFTER the core (ZoeDepth head, fusion network, tiling at the
384x512 / multiple-of-32 geometry, `r<N>` random tiles) is pinned by giving the oracle and the engine the SAME stand-in
through the reference's own injection point (ZoeDepth.forward(hack_feature=...), zoedepth_v1.py:160-166).
The stand-in depends on the image content (pooled colours through fixed random projections), so crops matter."""
import torch
import torch.nn.functional as F


class StandInCore:
    def __init__(self, seed, channels=256):
        g = torch.Generator().manual_seed(seed)
        self.w = [torch.randn(channels, 3, 1, 1, generator=g) * 1.5 for _ in range(5)]       # btlnck (/32), blocks /16 ... /2
        self.b = [torch.randn(channels, generator=g) * 0.3 for _ in range(5)]
        self.w_out = torch.randn(32, 3, 1, 1, generator=g)
        self.w_rel = torch.tensor([0.6, 1.1, 0.4]).view(1, 3, 1, 1)

    def __call__(self, x):
        x = x.float()
        dev = x.device
        feats = []
        for w, b, s in zip(self.w, self.b, (32, 16, 8, 4, 2)):
            feats.append(torch.tanh(F.conv2d(F.avg_pool2d(x, s), w.to(dev), b.to(dev))))
        feats.append(F.relu(F.conv2d(x, self.w_out.to(dev))))
        rel = F.conv2d(x, self.w_rel.to(dev))[:, 0] * 2.0
        return rel, feats
