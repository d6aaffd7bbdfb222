"""GPU probe: timing decomposition of the split GEMM with the debug library (PF_LIB_PATH=patchfusion_amd/libpf_wfdbg.so): PF_S3_DBG bit 0 =
no DMA after the ring fill, bit 1 = no MFMA, bit 2 = no fragment reads.  Results are wrong by construction; only the times mean anything."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk       # noqa: E402
from patchfusion_amd.hip_ops import ops        # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8296
g = torch.Generator().manual_seed(0)
for K, N in [(1024, 3072), (4096, 1024)]:
    w = torch.randn(N, K, generator=g) / K ** 0.5
    pw3 = pk.pack_conv_split3(w, torch.zeros(N)).to("cuda")
    x3 = torch.randn(3, M, K, generator=g).to(torch.bfloat16).to("cuda")
    y = torch.zeros(M, N, device="cuda")
    for tile in ("128", "64"):
        os.environ["PF_S3_TILE_NOW"] = tile
        row = []
        for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
            os.environ["PF_S3_DBG"] = str(dbg)
            t = ops.conv_split3(x3, pw3, y, _timed=10)
            row.append(f"dbg{dbg}={t * 1e3:.0f}us")
        print(f"M={M} K={K} N={N} tile={tile}: " + "  ".join(row), flush=True)
