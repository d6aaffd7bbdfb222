"""GPU probe: the four ViT-L block linears at the pass's token counts through the engine's entry (ops.conv_split3), interleaved over environment variants (an untimed
pass, then R rounds in which every variant is timed once).   usage: python tools/vit_linear_time.py "PF_S3_BALANCE=0" "" ...   (profiles/r6_balanced_walk.md)"""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd import packing as pk
from patchfusion_amd.hip_ops import ops
variants = sys.argv[1:] or [""]
DEV = "cuda"
g = torch.Generator().manual_seed(0)
print("| linear | M | " + " | ".join(f"`{v or 'default'}` ms (of 416.7)" for v in variants) + " |")
print("|---|---|" + "---|" * len(variants))
for M in (8 * 1037, 1037):
    for name, K, N, act, res, scale, split_out in (("qkv", 1024, 3072, None, False, False, True), ("proj", 1024, 1024, None, True, True, False),
                                                   ("fc1", 1024, 4096, "gelu", False, False, True), ("fc2", 4096, 1024, None, True, True, False)):
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        sc = (0.5 + torch.rand(N, generator=g)) if scale else None
        pw3 = pk.pack_conv_split3(w, b, scale=sc).to(DEV)
        x3 = torch.randn(3, K // 32, M, 32, generator=g).to(torch.bfloat16).to(DEV)       # chunk-major planes, as the producers write them
        r = torch.randn(M, N, generator=g).to(DEV) if res else None
        y = torch.empty(3, N // 32, M, 32, dtype=torch.bfloat16, device=DEV) if split_out else torch.empty(M, N, device=DEV)
        best = [1e9] * len(variants)
        for rnd in range(4):
            for i, v in enumerate(variants):
                kv = dict(x.split("=", 1) for x in v.split(",") if x)
                os.environ.update(kv)
                t = ops.conv_split3(x3, pw3, y, act=act, res=r, _timed=20)
                for k in kv:
                    os.environ.pop(k)
                if rnd:
                    best[i] = min(best[i], t)
        fl = 2.0 * M * K * N
        print(f"| {name} {K}->{N} | {M} | " + " | ".join(f"{t:.4f} ({fl / t / 1e9 / 416.7:.3f})" for t in best) + " |")
