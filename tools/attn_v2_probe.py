"""GPU probe of the version-2 split attention (csrc/attn_split3.hip), B = 8 x 16 heads x 1037 tokens, 32 queries per wave.
  decomp : timing decomposition on the ADBG build (make -C patchfusion_amd/csrc attndbg; PF_LIB_PATH=patchfusion_amd/libpf_attndbg.so): the launch with parts switched
           off (bit 1 no DMA in the loop, 2 no softmax VALU, 4 no QK MFMAs, 8 no PV, 16 no barriers); interleaved rounds, best of R
  launch [n] [qw] [B]: n plain launches (for rocprofv3 --pmc / --stats; tools/gpu_session.sh-style passes in tools/attn_pmc.sh)
usage: python tools/attn_v2_probe.py decomp | launch 12"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd.hip_ops import _L, _p, _stream, check, ops
mode = sys.argv[1]
S, H = 1037, 16
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
qw = int(sys.argv[3]) if len(sys.argv) > 3 else 32
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * S, 3 * H * 64, generator=g).cuda() * 0.5
q3 = torch.empty(3, B * S, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
ops.split3(qkv, q3)                                      # real h / m / l planes of float32 values (what the pass sees)
out = torch.empty(3, H * 64 // 32, B * S, 32, dtype=torch.bfloat16, device="cuda")
run = lambda: check(_L.pf_vit_attention_split3_v2(_p(q3), q3.stride(0), _p(out), out.stride(0), 1, B, S, H, qw, int(os.environ.get('PF_ATTN_SCHED', '0')), _stream()), "v2")
if mode == "launch":
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        run()
    torch.cuda.synchronize()
    sys.exit(0)
masks = [0, 1, 2, 4, 8, 16, 17, 2 | 16, 4 | 8, 2 | 4 | 8, 1 | 2 | 4 | 8, 31, 4 | 2, 8 | 2, 1 | 2, 0]
names = {1: "noDMA", 2: "noSoftmax", 4: "noQK", 8: "noPV", 16: "noBarrier"}
times = {i: [] for i in range(len(masks))}
for i, m in enumerate(masks):
    os.environ["PF_ATTN_DBG"] = str(m)
    for _ in range(3):
        run()
torch.cuda.synchronize()
for _ in range(3):
    for i, m in enumerate(masks):
        os.environ["PF_ATTN_DBG"] = str(m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            run()
        e1.record(); e1.synchronize()
        times[i].append(e0.elapsed_time(e1) / 30 * 1e3)
for i, m in enumerate(masks):
    label = "+".join(v for k, v in names.items() if m & k) or "full"
    print(f"dbg {m:2d} {label:40s}: best {min(times[i]):7.1f} us  ({' '.join(f'{x:.1f}' for x in times[i])})")
