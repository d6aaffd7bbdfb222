"""Per-op, per-shape roofline table of EVERY kernel launch of one 4K image pass.

Runs the bench workload (Depth-Anything ViT-L, 2160x3840, 4x4 tiles, process_num 8) once with a recorder around
every HipOps entry point, keeps the first occurrence of every distinct (op, shapes, dtypes, strides, scalars) call with
its real argument tensors, then times each distinct call standalone (torch.cuda events on the launch stream; the
kernels are launched on torch's current stream) and prices it against the roofline that bounds it:
  * conv / linear / attention  -> the launch's OWN useful multiply-adds / time  vs the peak of the pipe it EXECUTES on: the f32 MFMA
                                  (157.3 TF/s) for the f32-MFMA kernels, 2500 / 6 = 416.7 TF/s for the split-precision kernels (six bf16 MFMAs
                                  per float32-grade product), 2.5 PF/s in bf16 mode.  Winograd layers are priced on their transform-domain
                                  multiply-adds (a quarter of the direct convolution's) over the WHOLE layer time (transforms included); the
                                  direct-convolution equivalent is printed beside it.  No row can exceed 100 %.
  * everything else            -> algorithmic bytes (every logical input element read once + every output element
                                  written once; ROI ops: only the ROI region of the source) / time  vs 8 TB/s HBM

usage: python tools/op_roofline.py <bf16|fp32> [out.md] [out.json]
"""
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

try:
    from patchfusion_amd import hip_ops  # noqa: E402
except Exception:  # dry run on a box without the library
    hip_ops = None
from patchfusion_amd.config import make_config  # noqa: E402
from patchfusion_amd.model import PatchFusion  # noqa: E402
from patchfusion_amd.packing import PackedConv, winograd_applies  # noqa: E402
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict  # noqa: E402

HBM_PEAK = 8.0e12
MFMA_PEAK = {"bf16": 2500e12, "fp32": 157.3e12}
SPLIT_PEAK = 2500e12 / 6.0          # float32-grade products per second on the bf16 MFMA (six instructions per product)


def _as4(t):
    if t.dim() == 2:
        return t.unsqueeze(0).unsqueeze(0)
    if t.dim() == 3:
        return t.unsqueeze(0)
    return t


def nbytes(t, channels=None):
    """logical bytes of a (possibly channel-sliced) NHWC view"""
    if t is None:
        return 0
    n = t.numel() if channels is None else t.numel() // t.shape[-1] * channels
    return n * t.element_size()


def sig(v):
    if isinstance(v, torch.Tensor):
        return ("T", tuple(v.shape), str(v.dtype).replace("torch.", ""), tuple(v.stride()))
    if isinstance(v, PackedConv):
        return ("W", v.cin, v.cout, v.KH, v.KW, v.shuffle, tuple(v.w.shape))
    if isinstance(v, (tuple, list)):
        return tuple(sig(x) for x in v)
    if type(v).__name__ == "BinsTail":
        return ("BT", v.nq)
    return v


def work(name, a, k):
    """-> (kind, amount, description): kind 'flop' or 'byte'"""
    if name == "conv":
        x, pw, y = a[0], a[1], a[2]
        x4, y4 = _as4(x), _as4(y)
        s = max(pw.shuffle, 1)
        opix = y4.shape[0] * (y4.shape[1] // s) * (y4.shape[2] // s)
        fl = 2.0 * opix * pw.cin * pw.KH * pw.KW * pw.cout
        wino = winograd_applies(pw, x4.shape[0] * x4.shape[1] * x4.shape[2], k.get("stride", 1), k.get("pad", 0), k.get("act"))
        from patchfusion_amd import hip_ops
        fused = wino and pw.wino_up is not None and hip_ops._fused_wanted(x4.shape[0], x4.shape[1], x4.shape[2], pw)
        tag = ""
        desc = f"{tuple(x4.shape[:3])} {pw.cin}->{pw.cout} k{pw.KH} s{k.get('stride', 1)}" + (" convT" if s > 1 else "")
        if wino and (fused or pw.wino_u is not None):
            m = pw.wino_m
            own = fl * (m + 2) ** 2 / (9.0 * m * m)          # the layer's own multiply-adds in the transform domain (tile padding not counted)
            split = (not fused) and m == 4 and hip_ops._split3_three_step(pw)
            tag = (f" [winograd F{m} FUSED kernel, f32 MFMA]" if fused else
                   f" [winograd F{m}: 3 steps, {'split-precision GEMM on the bf16 MFMA' if split else 'f32 MFMA GEMM'}; time incl. transforms]")
            return "flop", own, desc + tag + f" (direct-conv equivalent x{fl / own:.2f})", (SPLIT_PEAK if split else None)
        if (pw.w3 is not None and x4.dtype == torch.float32 and pw.KH == 1 and pw.KW == 1 and k.get("stride", 1) == 1 and k.get("pad", 0) == 0 and s == 1 and
                hip_ops._conv1x1_split3_wanted(x4.shape[0] * x4.shape[1] * x4.shape[2], pw)):
            # float32 1x1 layer on the bf16 matrix cores, activations split in the kernel (csrc/conv1x1_split3.hip)
            return "flop", fl, desc + " [split-precision bf16x3, float32 in / out]", SPLIT_PEAK
        return "flop", fl, desc
    if name == "conv_split3":
        x3, pw, y = a[0], a[1], a[2]
        rows = x3.shape[2] if x3.dim() == 4 else x3.shape[1]          # chunk-major planes are [3, K/32, M, 32]
        fl = 2.0 * rows * pw.cin * pw.cout
        return "flop", fl, (f"(1, 1, {rows}) {pw.cin}->{pw.cout} k1 s1 [split-precision bf16x3]" + (" -> planes" if y.dtype == torch.bfloat16 else "")), SPLIT_PEAK
    if name == "layernorm_split3":
        x, y3 = a[0], a[1]
        return "byte", nbytes(x) + nbytes(y3), f"rows {x.shape[0]} D{x.shape[1]} -> three bf16 planes"
    if name == "split3":
        return "byte", nbytes(a[0]) + nbytes(a[1]), f"{tuple(a[0].shape)} -> three bf16 planes"
    if name == "vit_attention":
        qkv, out, B, S, heads = a[:5]
        if qkv.dim() == 3:
            return "flop", 4.0 * B * heads * S * S * 64, f"B{B} S{S} heads{heads} [split-precision bf16x3, planes in / out]", SPLIT_PEAK
        return "flop", 4.0 * B * heads * S * S * 64, f"B{B} S{S} heads{heads}" + (" (reads the QKV rows, no split)" if qkv.dtype == torch.float32 else " (incl. qkv_split)") + \
            (" -> planes" if qkv.dtype == torch.float32 and out.dtype == torch.bfloat16 else "")
    if name == "swin_window_attention":
        qkv, out = a[0], a[1]
        return "byte", nbytes(qkv) + nbytes(out), f"tokens {qkv.shape[0]} C{out.shape[1]} heads{a[7]}"
    if name == "patch_im2col":
        return "byte", nbytes(a[0]) + nbytes(a[1], 588), f"{tuple(a[0].shape)}"
    if name == "assemble_tokens":
        return "byte", nbytes(a[0]) + nbytes(a[1]) + nbytes(a[3]), f"{tuple(a[1].shape)}"
    if name == "layernorm":
        x, y = a[0], a[1]
        return "byte", nbytes(y) / y.element_size() * x.element_size() + nbytes(y), f"rows {y.numel() // y.shape[-1]} D{y.shape[-1]}"
    if name == "swin_ln_partition":
        return "byte", nbytes(a[0]) + nbytes(a[1]), f"{tuple(a[0].shape)}"
    if name == "swin_unpartition_add":
        return "byte", nbytes(a[1]) * 2 + nbytes(a[2]), f"{tuple(a[1].shape)}"
    if name == "add_rowwise":
        return "byte", 2 * nbytes(a[0]) + nbytes(a[1]), f"{tuple(a[0].shape)}"
    if name == "resize":
        x, y = a[0], a[1]
        add = k.get("add", a[2] if len(a) > 2 else None)
        return "byte", nbytes(x) + nbytes(y) + nbytes(add), f"{tuple(_as4(x).shape)} -> {tuple(_as4(y).shape[1:3])}"
    if name == "resize_concat":
        xs, y = a[0], a[1]
        return "byte", sum(nbytes(x) for x in xs) + nbytes(y, sum(x.shape[-1] for x in xs)), \
            f"{'+'.join(str(x.shape[-1]) for x in xs)} ch from {tuple(xs[-1].shape[1:3])} -> {tuple(_as4(y).shape[1:3])}"
    if name == "silog_loss":
        return "byte", nbytes(a[0]) + nbytes(a[1]), f"{tuple(a[0].shape)}"
    if name == "crop_resize":
        img, boxes, out = a[:3]
        b = boxes.cpu()
        area = int(((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).sum())
        return "byte", area * 3 * 4 + nbytes(out), f"{out.shape[0]} crops -> {tuple(out.shape[2:])}"
    if name in ("roi_align", "roi_align_depth"):
        feat, rois, y, scale = a[:4]
        r = rois.cpu().float()
        area = float(((r[:, 3] - r[:, 1]) * (r[:, 4] - r[:, 2])).clamp(min=0).sum()) * scale * scale
        per_px = nbytes(feat) / (feat.shape[-2] * feat.shape[-1] if name == "roi_align_depth" else feat.shape[1] * feat.shape[2])
        return "byte", area * per_px + nbytes(y), f"{tuple(feat.shape)} -> {tuple(y.shape)}"
    if name in ("maxpool2", "copy_channels"):
        return "byte", nbytes(a[0]) + nbytes(a[1], a[0].shape[-1]), f"{tuple(a[0].shape)}"
    if name == "pack_fusion_input":
        return "byte", nbytes(a[0]) + nbytes(a[1]) + nbytes(a[2]) + nbytes(a[3]), f"{tuple(a[3].shape)}"
    if name == "copy_plane":
        return "byte", 2 * nbytes(a[1]), f"{tuple(a[1].shape)}"
    if name == "nhwc_to_nchw":
        return "byte", nbytes(a[0]) + a[0].numel() * 4, f"{tuple(a[0].shape)}"
    if name == "attractor":
        A, n_attr, b_prev, out = a[:4]
        return "byte", nbytes(A, n_attr) + nbytes(b_prev) + nbytes(out), f"{tuple(out.shape)} n_attr{n_attr}"
    if name == "bins_tail":
        clb, emb, tw, cen, depth = a[:5]
        ct = 160 if tw.nq == 10 else 168
        fl = 2.0 * depth.numel() * (ct * 80 + 80 * 4)
        return "flop", fl, f"{tuple(depth.shape)} [{ct}->80->4 + log-binomial, embedding / centres from {tuple(emb.shape[1:3])}; rate = the two layers' FLOPs / time]"
    if name == "logbinom_depth":
        pt, centers, depth = a[:3]
        return "byte", nbytes(pt, 4) + nbytes(centers) + nbytes(depth), f"{tuple(depth.shape)} centres {tuple(centers.shape[1:3])}"
    if name == "stitch_init":
        return "byte", nbytes(a[2]) * 3 + nbytes(a[3]), f"{tuple(a[2].shape)} into {tuple(a[0].shape)}"
    if name == "stitch_finish_init":
        return "byte", 3 * nbytes(a[0]), f"{tuple(a[0].shape)}"
    if name == "stitch_update":
        return "byte", 5 * nbytes(a[2]) + nbytes(a[3]), f"{tuple(a[2].shape)}"
    if name in ("resize_nearest_f32", "resize_bilinear_f32"):
        return "byte", nbytes(a[0]) + nbytes(a[1]), f"{tuple(a[0].shape)} -> {tuple(a[1].shape)}"
    return None


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    fake = os.environ.get("PF_OP_ROOFLINE_DRYRUN") == "1"      # CPU dry run of this script's bookkeeping (tests/fake_ops, tiny config)
    if fake:
        from tests.fake_ops import ops as fops
        dev = torch.device("cpu")
        raw, pn = (448, 616), 2
        cfg = make_config("vits", (112, 154), raw, (2, 2))
        m = PatchFusion(cfg, compute_dtype="fp32", ops=fops).eval()
        sync = lambda: None
    else:
        dev = torch.device("cuda", 0)
        raw, pn = (2160, 3840), 8
        cfg = make_config("vitl", (392, 518), raw, (4, 4))
        m = PatchFusion(cfg, compute_dtype=dtype).eval()
        sync = torch.cuda.synchronize
    m.load_state_dict(synthetic_state_dict(patchfusion_spec(cfg), 0), strict=True)
    m = m.to(dev)
    m.overlap_coarse = m.overlap_batches = False
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234)).to(dev)
    lr = m.resizer(img)
    with torch.no_grad():
        m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=pn)        # warm-up (allocator, LDS attributes)
    sync()

    calls, first = collections.Counter(), {}
    ops = m.ops
    names = [n for n in dir(type(ops)) if not n.startswith("_") and n not in ("empty", "zeros", "name")]
    orig = {n: getattr(ops, n) for n in names}

    def wrap(n):
        f = orig[n]

        def g(*a, **k):
            key = (n, sig(a), sig(tuple(sorted(k.items()))))
            calls[key] += 1
            if key not in first:
                first[key] = (a, k)
            return f(*a, **k)
        return g

    for n in names:
        setattr(ops, n, wrap(n))
    with torch.no_grad():
        m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=pn)
    sync()
    for n in names:
        delattr(ops, n)           # instance attributes shadowed the class's static methods

    rows = []
    for key, (a, k) in first.items():
        n = key[0]
        w = work(n, a, k)
        if w is None:
            continue
        f = orig[n]
        for _ in range(2):
            f(*a, **k)
        iters = 5
        if fake:
            import time
            t0 = time.perf_counter()
            f(*a, **k)
            us = (time.perf_counter() - t0) * 1e6
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f(*a, **k)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
        kind, amount, desc = w[:3]
        peak = (w[3] if len(w) > 3 and w[3] else MFMA_PEAK[dtype]) if kind == "flop" else HBM_PEAK
        rate = amount / (us * 1e-6)
        frac = rate / peak
        rows.append(dict(op=n, shape=desc, launches=calls[key], us=us, kind=kind, amount=amount, rate=rate, peak=peak, frac=frac,
                         total_ms=calls[key] * us / 1e3))
    rows.sort(key=lambda r: -r["total_ms"])
    tot = sum(r["total_ms"] for r in rows)
    by_op = collections.defaultdict(float)
    for r in rows:
        by_op[(r["op"], r["kind"])] += r["total_ms"]
    out = [f"# every kernel launch of one 4K image pass (ViT-L, P=16, process_num=8, {dtype}), each distinct call timed standalone", "",
           f"sum of standalone times: {tot:.1f} ms per image; {sum(r['launches'] for r in rows)} launches, {len(rows)} distinct (op, shape) calls.",
           f"MFMA-bound rows: the launch's own useful TFLOP/s vs the peak of the pipe it executes on ({MFMA_PEAK[dtype] / 1e12:.1f} TF/s; split-precision rows "
           f"[bf16x3]: {SPLIT_PEAK / 1e12:.1f} TF/s = the bf16 MFMA peak / 6); HBM-bound rows: algorithmic GB/s vs 8000 GB/s.",
           "Rows tagged [winograd Fm ...] run the float32 Winograd layer (fused kernel csrc/wino_fused.hip, or the three steps of csrc/winograd.hip): priced on the layer's own "
           "transform-domain multiply-adds over the whole layer time; multiply by the printed factor for the direct-convolution equivalent.", "",
           "## per op", "", "| op | bound | total ms / image | share |", "|---|---|---:|---:|"]
    for (op, kind), t in sorted(by_op.items(), key=lambda kv: -kv[1]):
        out.append(f"| {op} | {'MFMA' if kind == 'flop' else 'HBM'} | {t:.2f} | {100 * t / tot:.1f}% |")
    out += ["", "## per (op, shape)", "", "| op | shape | launches | us / launch | rate | peak of its pipe | % of roofline | total ms |", "|---|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        rate = f"{r['rate'] / 1e12:.1f} TF/s" if r["kind"] == "flop" else f"{r['rate'] / 1e9:.0f} GB/s"
        pk = f"{r['peak'] / 1e12:.1f} TF/s" if r["kind"] == "flop" else "8000 GB/s"
        out.append(f"| {r['op']} | {r['shape']} | {r['launches']} | {r['us']:.1f} | {rate} | {pk} | {100 * r['frac']:.1f} | {r['total_ms']:.3f} |")
    txt = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    if len(sys.argv) > 3:
        json.dump(rows, open(sys.argv[3], "w"), indent=0)
    print("\n".join(out[:40]))


if __name__ == "__main__":
    main()
