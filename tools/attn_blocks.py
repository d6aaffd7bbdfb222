"""GPU probe (decomposition build, PF_LIB_PATH=patchfusion_amd/libpf_attndbg.so): the life of every block of one pipelined split-attention launch (B x 16 heads x 1037
tokens): entry / exit on the chip-wide 100 MHz clock, the CU it ran on.  Prints block time statistics, blocks in flight over time, blocks per CU.
usage: python tools/attn_blocks.py [B]"""
import ctypes as C, os, sys, collections, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from patchfusion_amd.hip_ops import _L, _p, _stream, check, ops
from patchfusion_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
S, H = 1037, 16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
qkv = torch.randn(B * S, 3 * H * 64).cuda() * 0.5
q3 = torch.empty(3, B * S, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
ops.split3(qkv, q3)
out = torch.empty(3, H * 64 // 32, B * S, 32, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    check(_L.pf_vit_attention_split3_v2(_p(q3), q3.stride(0), _p(out), out.stride(0), 1, B, S, H, 32, 2, _stream()), "v2")
torch.cuda.synchronize()
n = 9 * B * H
buf = (C.c_longlong * (4 * n))()
assert lib.pf_attn_dbg_blocks(buf, n) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n)]
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
print(f"B{B}: {n} blocks, launch span {(t1 - t0) / 100:.1f} us (100 MHz ticks)")
dur = sorted((r[1] - r[0]) / 100 for r in rows)
print(f"block time us: min {dur[0]:.1f} p10 {dur[n // 10]:.1f} median {dur[n // 2]:.1f} p90 {dur[9 * n // 10]:.1f} max {dur[-1]:.1f}; sum {sum(dur):.0f} us -> mean in flight {sum(dur) / ((t1 - t0) / 100):.0f}")
tail = [(r[1] - r[0]) / 100 for r in rows[:B * H]]
print(f"tail blocks (ids < {B * H}): mean {sum(tail) / len(tail):.1f} us; full blocks: mean {sum((r[1] - r[0]) / 100 for r in rows[B * H:]) / (n - B * H):.1f} us")
cu = collections.Counter()
for r in rows:
    hw = r[2]
    cu[(r[3] & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)] += 1
cnt = collections.Counter(cu.values())
print(f"distinct (xcc, se, sh, cu): {len(cu)}; blocks per CU histogram: {sorted(cnt.items())}")
steps = 20
for k in range(steps + 1):
    t = t0 + (t1 - t0) * k // steps
    print(f"  t = {(t - t0) / 100:6.1f} us: in flight {sum(1 for r in rows if r[0] <= t < r[1]):4d}   started {sum(1 for r in rows if r[0] <= t):4d}")
xc = collections.Counter((r[3] & 0xf) for r in rows)
print("blocks per XCC:", sorted(xc.items()))
starts = sorted(r[0] - t0 for r in rows)
print("start times (us) of blocks 0, 255, 256, 511, 512, 513, 600, 767, 768, 1023, last:", [f"{starts[i] / 100:.1f}" for i in (0, 255, 256, 511, 512, 513, 600, 767, 768, min(1023, n - 1), n - 1) if i < n])
