"""Interleaved A/B of the WHOLE image pass (BASELINE configs[2]: DA-vitl, 2160x3840, 4x4 m1, process_num 8, f32 headline dispatch) under
environment-switch variants, in ONE process on ONE box: round r runs every variant once (1 warm-up + K timed images each), so box-to-box and
DVFS drift (+-5 %) hits every arm alike.
usage: python tools/image_ab.py [--steps K] [--rounds R] [--split RxC] [--process-num N] "NAME=VAL,NAME2=VAL2" "NAME=VAL" ...
       the empty string "" is the default tree; model-level switches (PF_STREAMS, PF_VIT_BATCH_ALL, PF_OVERLAP_BATCHES) are applied to the live model.
Also prints max |depth - depth(variant 0)| so that a schedule-only switch is seen to leave the numbers alone."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--split", default="4x4")
    ap.add_argument("--process-num", type=int, default=8)
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    from patchfusion_amd import hip_ops
    from patchfusion_amd.config import make_config
    from patchfusion_amd.model import PatchFusion
    from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    split = tuple(int(v) for v in a.split.split("x"))
    cfg = make_config(a.encoder, (392, 518), (2160, 3840), split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).to(dev)
    model = PatchFusion(cfg, compute_dtype="fp32").eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    lr = model.resizer(img)
    defaults = dict(n_streams=model.n_streams, vit_batch_all=model.vit_batch_all, overlap_batches=model.overlap_batches)
    variants = [dict(kv.split("=", 1) for kv in v.split(",") if kv) for v in a.variants]
    touched = sorted({k for v in variants for k in v})

    def apply(v):
        for k in touched:
            os.environ.pop(k, None)
        os.environ.update(v)
        model.n_streams = int(v.get("PF_STREAMS", defaults["n_streams"]))
        model.vit_batch_all = v["PF_VIT_BATCH_ALL"] == "1" if "PF_VIT_BATCH_ALL" in v else defaults["vit_batch_all"]
        model.overlap_batches = v["PF_OVERLAP_BATCHES"] != "0" if "PF_OVERLAP_BATCHES" in v else defaults["overlap_batches"]
        hip_ops.refresh_env()

    def step():
        d, _ = model(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=a.process_num)
        return d

    times = [[] for _ in variants]
    ref = None
    diffs = [0.0] * len(variants)
    for r in range(a.rounds):
        for i, v in enumerate(variants):
            apply(v)
            d = step()
            torch.cuda.synchronize()
            if r == 0:
                if i == 0:
                    ref = d.clone()
                diffs[i] = float((d - ref).abs().max())
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / a.steps * 1e3)
            print(f"round {r} variant {i} {a.variants[i]!r}: {times[i][-1]:.2f} ms", file=sys.stderr, flush=True)
    print(f"| variant | " + " | ".join(f"round {r} ms" for r in range(a.rounds)) + " | mean ms | vs variant 0 | max abs depth diff vs variant 0 |")
    print("|---|" + "---|" * (a.rounds + 3))
    m0 = sum(times[0]) / len(times[0])
    for i, v in enumerate(a.variants):
        m = sum(times[i]) / len(times[i])
        print(f"| `{v or 'default'}` | " + " | ".join(f"{t:.2f}" for t in times[i]) + f" | {m:.2f} | {m - m0:+.2f} | {diffs[i]:.2e} |")
    print(f"\n(split {a.split}, process_num {a.process_num}, {a.steps} images per cell, peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")


if __name__ == "__main__":
    main()
