"""GPU probe: the split-precision ViT attention launch alone, version 1 (csrc/vit.hip vit_attention_split3_kernel) against version 2 (csrc/attn_split3.hip, 16 and
32 queries per wave), B = 8 and B = 1 x 16 heads x 1037 tokens, random bf16 planes in, chunk-major planes out.  Interleaved: an untimed pass, then R rounds in
which every variant is timed once over N launches with HIP events (no arm sits on a cold clock).  A/B against another build: PF_LIB_PATH=<other libpf_hip.so>.
usage: python tools/attn_split3_time.py [rounds]   (profiles/r6_attention_v2.md)"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd.hip_ops import _L, _p, _stream, check, ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = 40
g = torch.Generator().manual_seed(0)
S, H = 1037, 16
for B in (8, 1):
    qkv = torch.randn(B * S, 3 * H * 64, generator=g).cuda() * 0.5
    q3 = torch.empty(3, B * S, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
    ops.split3(qkv, q3)                                  # real h / m / l planes of float32 values (what the pass sees)
    out = torch.empty(3, H * 64 // 32, B * S, 32, dtype=torch.bfloat16, device="cuda")
    variants = {"v1": lambda: check(_L.pf_vit_attention_split3(_p(q3), q3.stride(0), _p(out), out.stride(0), 1, B, S, H, _stream()), "v1")}
    for sched, qws in ((1, (16, 32)), (2, (32,))):
        for qw in qws:
            variants[f"v2_s{sched}_qw{qw}"] = (lambda qw=qw, sched=sched: check(_L.pf_vit_attention_split3_v2(_p(q3), q3.stride(0), _p(out), out.stride(0), 1, B, S, H, qw,
                                                                                                                sched, _stream()), "v2"))
    for f in variants.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    times = {k: [] for k in variants}
    for _ in range(R):
        for k, f in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(N):
                f()
            e1.record(); e1.synchronize()
            times[k].append(e0.elapsed_time(e1) / N * 1e3)
    for k, t in times.items():
        us = min(t)
        tf = 4.0 * B * H * S * S * 64 / us / 1e6
        print(f"B{B} {k:12s}: best {us:7.1f} us (rounds {' '.join(f'{x:.1f}' for x in t)}) = {tf:6.1f} TF/s useful = {tf / 416.7:.3f} of 2500/6")
