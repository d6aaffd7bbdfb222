"""Measure (not assert) the numerical error of the engine at BASELINE.json configs[2] itself: Depth-Anything ViT-L,
2160x3840, 4x4 tiles, process_num 8.

  python tools/precision_probe.py [--out gpurun_out/r2_precision_probe.json] [--skip-oracle]

1. final stitched depth of the f32 (exact) and bf16 (fast) engines against the ORACLE (oracle/pf_oracle.py, the CPU
   restatement pinned to the reference) evaluated with torch on the same GPU: max / p99.9 / p99 / mean |delta| in depth units;
2. per-stage error growth of the bf16 engine against the f32 engine (same kernels, other precision) on the coarse branch and
   one fusion batch: relative rms and max error of every tap -- this locates the stage that amplifies.
TEST INFRASTRUCTURE / measurement only (imports oracle/).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats(d, ref):
    diff = (d.float() - ref.float()).abs().flatten()
    sub = diff[::5]
    return {"max_abs": float(diff.max()), "p999_abs": float(torch.quantile(sub, 0.999)), "p99_abs": float(torch.quantile(sub, 0.99)),
            "mean_abs": float(diff.mean()), "ref_std": float(ref.float().std()), "ref_min": float(ref.min()), "ref_max": float(ref.max())}


def rel(a, b):
    a, b = a.float(), b.float()
    rms = float(b.pow(2).mean().sqrt()) + 1e-30
    d = (a - b)
    return {"rel_rms": float(d.pow(2).mean().sqrt()) / rms, "rel_max": float(d.abs().max()) / rms, "ref_rms": rms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r2_precision_probe.json"))
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--encoder", default="vitl")
    args = ap.parse_args()
    from oracle import pf_oracle
    from patchfusion_amd.config import make_config
    from patchfusion_amd.model import PatchFusion
    from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict

    raw, split, pn = (2160, 3840), (4, 4), 8
    cfg = make_config(args.encoder, (392, 518), raw, split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234)).cuda()
    out = {"config": f"DA-{args.encoder} 2160x3840 4x4 m1 process_num=8 (BASELINE configs[2])"}

    models, depth, taps = {}, {}, {}
    for dt in ("fp32", "bf16"):
        m = PatchFusion(cfg, compute_dtype=dt).eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        m.overlap_coarse = m.overlap_batches = False
        lr = m.resizer(img)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=pn)
        torch.cuda.synchronize()
        depth[dt] = d.float().clone()
        # stage taps: coarse branch (+G2L) and the first fusion batch
        t = {}
        st = m._coarse(lr, t)
        for i, f in enumerate(st["feats"]):
            t[f"coarse_feat{i}"] = f
        for i, f in enumerate(st["g2l"]):
            t[f"g2l{i}"] = f
        t["coarse_depth"] = st["depth"]
        from patchfusion_amd import tiling
        tiles = tiling.tile_schedule(m.tile_cfg, m.patch_process_shape, "m1", pn)
        bt, rois = m._tile_tables(tiles, m.tile_cfg)
        crops = torch.empty((pn, 3, 392, 518), dtype=torch.float32, device="cuda")
        m.ops.crop_resize(img[0].contiguous(), bt[:pn], crops)
        ft = {}
        fd, ff = m._engine["fine"].forward(m.ops, crops, ft)
        for k, v in ft.items():
            t["fine_" + k] = v
        for i, f in enumerate(ff):
            t[f"fine_feat{i}"] = f
        t["fine_depth"] = fd
        fu = {}
        dd = m._engine["fusion"].forward(m.ops, crops, rois[:pn], fd, ff, st["depth"], st["feats"], st["g2l"], fu)
        for k, v in fu.items():
            t["fusion_" + k] = v
        t["fusion_depth"] = dd
        torch.cuda.synchronize()
        taps[dt] = {k: v.float().clone() for k, v in t.items()}
        del m
        torch.cuda.empty_cache()

    out["bf16_vs_f32_engine_final"] = stats(depth["bf16"], depth["fp32"])
    out["bf16_vs_f32_engine_stages"] = {k: rel(taps["bf16"][k], taps["fp32"][k]) for k in taps["fp32"] if k in taps["bf16"]}
    for k in ("coarse_depth", "fine_depth", "fusion_depth"):
        out["bf16_vs_f32_engine_stages"][k].update(stats(taps["bf16"][k], taps["fp32"][k]))
    print("== bf16 engine vs f32 engine, per stage (rel_rms, rel_max) ==")
    for k, v in out["bf16_vs_f32_engine_stages"].items():
        print(f"  {k:28s} rel_rms {v['rel_rms']:.3e}  rel_max {v['rel_max']:.3e}  ref_rms {v['ref_rms']:.3e}")
    print("final:", out["bf16_vs_f32_engine_final"])

    if not args.skip_oracle:
        t0 = time.time()
        sdg = {k: v.cuda() for k, v in sd.items()}
        orc = pf_oracle.Oracle(cfg, sdg)
        lr = orc.resizer(img)
        ref = orc.infer(lr, img, "m1", pn)
        torch.cuda.synchronize()
        out["oracle_gpu_seconds"] = time.time() - t0
        out["f32_vs_oracle"] = stats(depth["fp32"], ref)
        out["bf16_vs_oracle"] = stats(depth["bf16"], ref)
        print("f32 vs oracle :", out["f32_vs_oracle"])
        print("bf16 vs oracle:", out["bf16_vs_oracle"])
        print(f"oracle on GPU took {out['oracle_gpu_seconds']:.1f}s")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
