"""GPU probe (decomposition build: make -C patchfusion_amd/csrc attndbg; PF_LIB_PATH=patchfusion_amd/libpf_attndbg.so): s_memtime phase sums of the pipelined split
attention, wave 0 of the first / middle / last block of a B x 16 heads x 1037 tokens launch.   usage: python tools/attn_timeline.py [B]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd.hip_ops import _L, _p, _stream, check, ops
from patchfusion_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
S, H = 1037, 16
names = ["prologue wait", "landing wait + barrier", "DMA issue", "-", "rescale + first reads + softmax head", "-", "-", "-", "the four super-groups",
         "-", "final wait", "epilogue"]
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [8, 1]):
    qkv = torch.randn(B * S, 3 * H * 64).cuda() * 0.5
    q3 = torch.empty(3, B * S, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
    ops.split3(qkv, q3)
    out = torch.empty(3, H * 64 // 32, B * S, 32, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        check(_L.pf_vit_attention_split3_v2(_p(q3), q3.stride(0), _p(out), out.stride(0), 1, B, S, H, 32, 2, _stream()), "v2")
    torch.cuda.synchronize()
    buf = (C.c_longlong * 48)()
    assert lib.pf_attn_dbg_timeline(buf) == 0
    print(f"== B{B}: cycles of wave 0 (s_memtime), blocks first / middle / last; 33 key blocks = 35 steps")
    for k, nm in enumerate(names):
        print(f"  [{k:2d}] {nm:38s}: " + "  ".join(f"{buf[b * 16 + k]:9d}" for b in range(3)))
    print("  total                          : " + "  ".join(f"{sum(buf[b * 16 + k] for k in range(12)):9d}" for b in range(3)))
