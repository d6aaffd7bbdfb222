"""Time two conv shapes with whatever library PF_LIB_PATH points at: the compile-time ablation builds of igemm.hip
(`make -C patchfusion_amd/csrc ablate`, or hipcc -DPF_ABL_NODMA / -DPF_ABL_NOBAR / -DPF_ABL_NOLDS on igemm.hip; their results
are wrong by construction, only the time matters).  Round-2 numbers: profiles/r2_abl_f32.log.
usage: PF_LIB_PATH=... [PF_IGEMM_CFG=1] python tools/abl_run.py <fp32|bf16>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import packing as pk  # noqa: E402
from patchfusion_amd.hip_ops import ops  # noqa: E402

dt = torch.float32 if sys.argv[1] == "fp32" else torch.bfloat16
dev = torch.device("cuda", 0)
for name, (B, H, W), cin, cout, k in [("up4_1", (8, 224, 296), 768, 768, 3), ("qkv", (1, 1, 8296), 1024, 3072, 1)]:
    x = torch.randn(B, H, W, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    pw = pk.pack_conv(w, torch.zeros(cout), dtype=dt).to(dev)
    y = torch.empty(B, H, W, cout, device=dev, dtype=dt)
    ms = ops.conv(x, pw, y, pad=k // 2, _timed=3)
    fl = 2.0 * B * H * W * cin * k * k * cout
    print(f"{os.path.basename(os.environ.get('PF_LIB_PATH', 'libpf_hip.so')):28s} {name:6s} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s", flush=True)
