"""GPU probe: the float32 1x1 convolutions / linears of one 4K image pass (configs[2], ViT-L) through the split-precision kernel with the in-loader
split (csrc/conv1x1_split3.hip) and through the f32-MFMA implicit GEMM (`_direct`), interleaved in one process (an untimed pass, then three rounds, best
of each).  Useful TF/s = 2 M K N / time; `of 416.7` = the bf16 peak / 6.  usage: python tools/conv1x1_time.py   (profiles/r6_conv1x1_split3.md)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PF_CONV1X1_SPLIT3", "2")
from patchfusion_amd import packing as pk        # noqa: E402
from patchfusion_amd.hip_ops import ops          # noqa: E402

# (tokens, Cin, Cout, act, launches per image)
SHAPES = ((8 * 224 * 296, 256, 128, "relu", 4), (8 * 224 * 296, 256, 256, None, 2), (8 * 224 * 296, 128, 128, "relu", 8), (66304, 256, 1024, "gelu", 2),
          (66304, 1024, 256, None, 2), (68400, 256, 768, None, 2), (68400, 256, 256, None, 2), (8 * 28 * 37, 1024, 1024, None, 4),
          (16576, 256, 1024, "gelu", 3), (16576, 1024, 256, None, 3), (18720, 256, 768, None, 3), (8 * 112 * 148, 256, 128, "relu", 4),
          (8 * 112 * 148, 256, 256, None, 2), (8 * 112 * 148, 128, 128, "relu", 8), (8 * 28 * 37, 1024, 512, None, 2), (8 * 56 * 74, 256, 128, "relu", 4),
          (203056, 32, 96, None, 2), (203056, 32, 128, "gelu", 2), (203056, 128, 32, None, 2), (8 * 392 * 518, 32, 32, None, 2))
g = torch.Generator().manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "launch":
    # N launches of ONE layer through the split route (tools/kernel_pmc.sh): python tools/conv1x1_time.py launch N tokens Cin Cout
    n, M, K, N = (int(v) for v in sys.argv[2:6])
    w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
    pw = pk.pack_conv(w, torch.randn(N, generator=g), dtype=torch.float32).to("cuda")
    x = torch.randn(1, 1, M, K, generator=g).to("cuda")
    y = torch.empty(1, 1, M, N, device="cuda")
    for _ in range(n):
        ops.conv(x, pw, y, act="relu", _direct=False)
    torch.cuda.synchronize()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "decomp":
    # timing decomposition with the PF_C1_DBG build (make -C patchfusion_amd/csrc variant NAME=c1dbg DEFS=-DPF_C1_DBG; PF_LIB_PATH=.../libpf_c1dbg.so):
    # parts of the kernel switched off one at a time (results wrong by construction)
    names = {0: "full", 1: "no MFMA", 2: "no token loads", 4: "no weight DMA", 8: "no split / plane writes", 16: "no fragment reads (one stage re-read)",
             32: "no barrier", 6: "no loads at all", 14: "no loads, no split", 15: "MFMA off + no loads + no split", 30: "only MFMA + barrier (no loads, split, frag re-reads)"}
    print("| layer | " + " | ".join(names.values()) + " |")
    print("|---|" + "---|" * len(names))
    for (M, K, N, act, n) in SHAPES[:2] + SHAPES[4:5]:
        w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
        pw = pk.pack_conv(w, torch.randn(N, generator=g), dtype=torch.float32).to("cuda")
        x = torch.randn(1, 1, M, K, generator=g).to("cuda")
        y = torch.empty(1, 1, M, N, device="cuda")
        row = []
        for bits in names:
            os.environ["PF_C1_DBG"] = str(bits)
            best = 1e9
            for rnd in range(3):
                t = ops.conv(x, pw, y, act=act, _timed=10, _direct=False)
                if rnd:
                    best = min(best, t)
            row.append(best * 1e3)
        print(f"| {M} x {K}->{N} | " + " | ".join(f"{t:.1f}" for t in row) + " |")
    sys.exit(0)
print("| tokens | layer | launches / image | f32 MFMA us (TF/s) | split us (TF/s, of 416.7) | speed-up | ms / image saved |")
print("|---|---|---|---|---|---|---|")
tot_a = tot_b = 0.0
for (M, K, N, act, n) in SHAPES:
    w = torch.randn(N, K, 1, 1, generator=g) / K ** 0.5
    pw = pk.pack_conv(w, torch.randn(N, generator=g), dtype=torch.float32).to("cuda")
    x = torch.randn(1, 1, M, K, generator=g).to("cuda")
    y = torch.empty(1, 1, M, N, device="cuda")
    best = [1e9, 1e9]
    for rnd in range(4):
        for i, direct in enumerate((True, False)):
            t = ops.conv(x, pw, y, act=act, _timed=10, _direct=direct)
            if rnd:
                best[i] = min(best[i], t)
    fl = 2.0 * M * K * N
    a, b = best[0] * 1e3, best[1] * 1e3
    tot_a += a * n
    tot_b += min(a, b) * n
    print(f"| {M} | {K}->{N} {act or ''} | {n} | {a:.1f} ({fl / a / 1e6:.0f}) | {b:.1f} ({fl / b / 1e6:.0f}, {fl / b / 1e6 / 416.7:.3f}) | {a / b:.2f} | {(a - b) * n / 1e3:.3f} |")
print(f"\nsum over one image: f32 MFMA {tot_a / 1e3:.2f} ms, with the split route where it wins {tot_b / 1e3:.2f} ms")
