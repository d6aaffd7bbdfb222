"""GPU debug probe: error of every split-attention variant against float64 on a few shapes, and where the pipelined kernel differs from the two-phase one."""
import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd.hip_ops import _L, _p, _stream, check, ops
g = torch.Generator().manual_seed(78)
for (B, S, heads, scale) in ((2, 1037, 16, 1.0), (1, 129, 1, 6.0), (1, 64, 1, 3.0), (1, 32, 1, 3.0), (1, 96, 1, 3.0), (1, 13, 2, 1.0), (9, 300, 6, 1.0), (8, 1037, 16, 0.5)):
    D = heads * 64
    qkv = (torch.randn(B * S, 3 * D, generator=g) * scale).cuda()
    q, k, v = qkv.double().view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * S, D)
    den = max(1.0, float(ref.abs().max()))
    q3 = torch.empty(3, B * S, 3 * D, dtype=torch.bfloat16, device="cuda")
    ops.split3(qkv, q3)
    outs = {}
    for qw, sched in ((32, 1), (32, 2)):
        o2 = torch.full((3, B * S, D), 7.0, dtype=torch.bfloat16, device="cuda")
        check(_L.pf_vit_attention_split3_v2(_p(q3), q3.stride(0), _p(o2), o2.stride(0), 0, B, S, heads, qw, sched, _stream()), "v2")
        outs[(qw, sched)] = o2.double().sum(0)
        e = (outs[(qw, sched)] - ref).abs()
        print(f"B{B} S{S} h{heads} x{scale} qw{qw} sched{sched}: max err {float(e.max()) / den:.2e} mean {float(e.mean()) / den:.2e}")
    d = (outs[(32, 2)] - outs[(32, 1)]).abs().view(B, S, heads, 64)
    print("   pipe - twophase max:", f"{float(d.max()):.1e}")
