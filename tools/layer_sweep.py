"""Per-layer roofline table of the conv / linear launches of ONE 4K image pass (kernel-tuning aid).

Runs the bench workload once with a recorder around HipOps.conv, groups the calls by shape, then times every
distinct shape standalone (pf_conv_timed: HIP events on the launch stream, the exact tensors' shapes / strides
rebuilt with random data) and prints a markdown table sorted by total standalone time:
    shape | launches per image | ms per launch | TFLOP/s | % of the 2.5 PF/s bf16 peak | total ms
usage: python tools/layer_sweep.py [out.md]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from patchfusion_amd import hip_ops, packing as pk  # noqa: E402
from patchfusion_amd.config import make_config  # noqa: E402
from patchfusion_amd.model import PatchFusion  # noqa: E402
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = make_config("vitl", (392, 518), (2160, 3840), (4, 4))
    m = PatchFusion(cfg, compute_dtype="bf16").eval()
    m.load_state_dict(synthetic_state_dict(patchfusion_spec(cfg), 0), strict=True)
    m = m.to(dev)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).to(dev)
    lr = m.resizer(img)
    calls = collections.Counter()
    orig = hip_ops.HipOps.conv

    def rec(x, pw, y, stride=1, pad=0, act=None, relu_in=False, res=None, res2=None, _timed=None):
        x4 = hip_ops._as4(x)
        key = (tuple(x4.shape[:3]), pw.cin, pw.cout, pw.KH, stride, pad, bool(relu_in), pw.shuffle, str(x4.dtype))
        calls[key] += 1
        return orig(x, pw, y, stride=stride, pad=pad, act=act, relu_in=relu_in, res=res, res2=res2, _timed=_timed)

    hip_ops.HipOps.conv = staticmethod(rec)
    m.ops.conv = rec
    with torch.no_grad():
        m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=8)
    torch.cuda.synchronize()
    hip_ops.HipOps.conv = staticmethod(orig)
    m.ops.conv = orig
    rows = []
    for key, n in calls.items():
        (B, H, W), cin, cout, k, stride, pad, relu_in, shuffle, dt = key
        if dt != "torch.bfloat16":
            continue
        x = torch.randn(B, H, W, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.bfloat16).to(dev)
        pw.shuffle = shuffle
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        s = max(shuffle, 1)
        y = torch.empty(B, OH * s, OW * s, cout // (s * s), device=dev, dtype=torch.bfloat16)
        ms = orig(x, pw, y, stride=stride, pad=pad, relu_in=relu_in, _timed=3)
        fl = 2.0 * B * OH * OW * cin * k * k * cout
        rows.append((n * ms, n, ms, fl / ms / 1e9, key))
        del x, y, pw
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    out = ["# conv / linear launches of one 4K image pass (ViT-L, P=16, process_num=8, bf16), each distinct shape timed standalone",
           "", f"sum of standalone times: {tot:.1f} ms per image ({sum(r[1] for r in rows)} launches, {len(rows)} distinct shapes)", "",
           "| input [B,H,W] | Cin→Cout | k/s | relu_in | launches | ms/launch | TFLOP/s | % of 2.5 PF/s | total ms | cum % |", "|---|---|---|---|---:|---:|---:|---:|---:|---:|"]
    cum = 0.0
    for t, n, ms, tf, key in rows:
        (B, H, W), cin, cout, k, stride, pad, relu_in, shuffle, dt = key
        cum += t
        out.append(f"| {B}×{H}×{W} | {cin}→{cout}{' (convT s%d)' % shuffle if shuffle > 1 else ''} | {k}/{stride} | {'y' if relu_in else ''} | {n} | {ms:.3f} | {tf:.0f} | "
                   f"{tf / 25:.1f} | {t:.2f} | {100 * cum / tot:.0f} |")
    txt = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
