"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REFERENCE'S OWN PYTHON
(/root/reference, via oracle/ref_shim.py) on seeded synthetic weights and inputs.

Run in the build container only:   python -m oracle.make_golden
The fixtures travel to the GPU box; the reference does not.

Cases (weights: patchfusion_amd.spec.synthetic_state_dict(seed=0); image: torch.rand with
Generator().manual_seed(1234); python `random` seeded 5621 before every forward, like
tools/test.py:122-123 fix_random_seed):
  tiny_vits : DA-vits, process 112x154, raw 448x616, split 2x2, process_num=2, modes m1 / m2 / r4,
              full output maps + coarse depth + sampled coarse features
  full_vits : DA-vits, process 392x518, raw 784x1036, split 2x2, process_num=4, mode m1, 8192
              sampled output values + stats (kept small)
  cfg4k_vits: BASELINE.json configs[0] and configs[1] themselves - DA-vits, 2160x3840 image:
              2x2 tiles, cai_mode r4, process_num=4 (13 patches; SURVEY 8d "Config 1") and
              4x4 tiles, cai_mode m1, process_num=4 (16 patches; "Config 2"); 16384 sampled values each
              (`python -m oracle.make_golden cfg4k` regenerates only this file; ~5 CPU-minutes)
  variants_vits: the tiny geometry with bin_centers_type / attractor_type / attractor_kind set to the values no shipped config
              uses (patchfusion.py:132-146, attractor.py:112-129), weights seed 3, image seed 11: full m1 map + coarse depth each
              (`python -m oracle.make_golden variants` regenerates only this file)
  headline_vitl_ref: the HEADLINE architecture (DA-vitl, process 392x518) from the reference itself: one 2160x3840 image, 2x2 tiles, m1,
              process_num=4; sampled final map + coarse depth + two coarse feature levels (`python -m oracle.make_golden vitl`; ~1-2 CPU-minutes)
  cfg2_vitl_ref / cfg3_vitl_ref: the BENCHED configurations themselves from the reference itself - BASELINE.json configs[2] (DA-vitl, 2160x3840,
              4x4 tiles, m1, process_num=8; 16 patches) and the configs[3] geometry (8x8 tiles, m1, process_num=8; 64 patches): the same four
              sampled tensors as headline_vitl_ref (`python -m oracle.make_golden cfg2` ~5 CPU-minutes, `... cfg3` ~20 CPU-minutes on 8 threads)
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from patchfusion_amd.config import make_config  # noqa: E402
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sample_idx(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (k,), generator=g).numpy()


def build(enc, ps, raw, split):
    PF = ref_shim.import_reference()
    cfg = make_config(enc, ps, raw, split)
    with ref_shim.in_reference_cwd():
        m = PF(cfg).eval()
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    print(m.load_state_dict(sd, strict=True))
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234))
    return m, cfg, img


def cfg4k():
    """BASELINE.json configs[0] / configs[1] at their real size, from the reference itself."""
    out = {}
    for name, split, mode in (("c0_2x2_r4", (2, 2), "r4"), ("c1_4x4_m1", (4, 4), "m1")):
        m, cfg, img = build("vits", (392, 518), (2160, 3840), split)
        lr = m.resizer(img)
        with torch.no_grad():
            random.seed(5621)
            d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=mode, process_num=4)
        flat = d.flatten()
        idx = sample_idx(flat.numel(), 16384, 11)
        out[name + "_shape"] = np.array(d.shape[2:], np.int64)
        out[name + "_idx"] = idx
        out[name + "_val"] = flat[idx].numpy()
        out[name + "_stats"] = np.array([d.mean().item(), d.std().item(), d.min().item(), d.max().item()], np.float32)
        print(name, tuple(d.shape), out[name + "_stats"], flush=True)
        del m
    np.savez_compressed(os.path.join(OUT, "cfg4k_vits.npz"), **out)


def vitl(split=(2, 2), process_num=4, fname="headline_vitl_ref.npz"):
    """(split, process_num, file) = ((2, 2), 4, headline_vitl_ref) | ((4, 4), 8, cfg2_vitl_ref) | ((8, 8), 8, cfg3_vitl_ref).
    The HEADLINE architecture from the reference itself (round-4 review, missing #5): DA-vitl, process 392x518, one 2160x3840 image cut in
    2x2 tiles, cai_mode m1, process_num 4 -- the reference's own PatchFusion.forward(mode='infer') (patchfusion.py:401-453; ViT-L widths
    depth_anything.py:346-352) on the seeded weights / image every headline test uses.  Stored: 8192 sampled values of the final map, 4096 of the
    coarse depth, 2048 of coarse feature level 3 (r2, 256 @ 112x148) and level 5 (out_conv, 32 @ 392x518), + stats.  ~1 CPU-minute on 8 threads."""
    m, cfg, img = build("vitl", (392, 518), (2160, 3840), split)
    lr = m.resizer(img)
    out = {}
    with torch.no_grad():
        random.seed(5621)
        cd, cf = m.coarse_forward(lr)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=process_num)
    for name, t, k, seed in (("depth_m1", d, 8192, 21), ("coarse_depth", cd, 4096, 22), ("coarse_feat3", cf[3], 2048, 23), ("coarse_feat5", cf[5], 2048, 24)):
        flat = t.flatten()
        idx = sample_idx(flat.numel(), k, seed)
        out[name + "_shape"] = np.array(t.shape, np.int64)
        out[name + "_idx"] = idx.astype(np.int32)
        out[name + "_val"] = flat[idx].numpy()
        out[name + "_stats"] = np.array([t.mean().item(), t.std().item(), t.min().item(), t.max().item()], np.float32)
        print(name, tuple(t.shape), out[name + "_stats"], flush=True)
    np.savez_compressed(os.path.join(OUT, fname), **out)


VARIANTS = (("normed", "inv", "mean"), ("hybrid1", "exp", "sum"), ("hybrid2", "inv", "sum"), ("softplus", "exp", "mean"))


def variant_case(kind, atype, akind):
    """config, synthetic weights and image of one variants_vits case (shared with the tests)"""
    cfg = make_config("vits", (112, 154), (448, 616), (2, 2))
    for b in ("coarse_branch", "fine_branch"):
        cfg[b].update(bin_centers_type=kind, attractor_type=atype, attractor_kind=akind)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 3)
    img = torch.rand(1, 3, 448, 616, generator=torch.Generator().manual_seed(11))
    return cfg, sd, img


def variants():
    PF = ref_shim.import_reference()
    out = {}
    for kind, atype, akind in VARIANTS:
        cfg, sd, img = variant_case(kind, atype, akind)
        with ref_shim.in_reference_cwd():
            m = PF(cfg).eval()
        m.load_state_dict(sd, strict=True)
        lr = m.resizer(img)
        with torch.no_grad():
            cd, _ = m.coarse_forward(lr)
            d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=2)
        out[f"{kind}_coarse_depth"] = cd[0, 0].numpy()
        out[f"{kind}_depth_m1"] = d[0, 0].numpy()
        print(kind, atype, akind, float(d.mean()), float(d.std()), float(d.min()), float(d.max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "variants_vits.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if len(sys.argv) > 1 and sys.argv[1] == "cfg4k":
        cfg4k()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        variants()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "vitl":
        vitl()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cfg2":
        vitl((4, 4), 8, "cfg2_vitl_ref.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3":
        vitl((8, 8), 8, "cfg3_vitl_ref.npz")
        return
    # ---------------- tiny ----------------
    m, cfg, img = build("vits", (112, 154), (448, 616), (2, 2))
    lr = m.resizer(img)
    out = {}
    with torch.no_grad():
        cd, cf = m.coarse_forward(lr)
        out["coarse_depth"] = cd.numpy()
        for i, f in enumerate(cf):
            flat = f.flatten()
            idx = sample_idx(flat.numel(), 2048, 100 + i)
            out[f"coarse_feat{i}_idx"] = idx
            out[f"coarse_feat{i}_val"] = flat[idx].numpy()
            out[f"coarse_feat{i}_stats"] = np.array([f.mean().item(), f.std().item(), f.abs().max().item()], np.float32)
        for mode in ("m1", "m2", "r4"):
            random.seed(5621)
            d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=mode, process_num=2)
            out[f"depth_{mode}"] = d[0, 0].numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_vits.npz"), **out)
    print("tiny_vits done", {k: v.shape for k, v in out.items() if k.startswith("depth")})
    # ---------------- full-size vits ----------------
    m, cfg, img = build("vits", (392, 518), (784, 1036), (2, 2))
    lr = m.resizer(img)
    out = {}
    with torch.no_grad():
        random.seed(5621)
        cd, cf = m.coarse_forward(lr)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=4)
    for name, t in (("coarse_depth", cd), ("depth_m1", d)):
        flat = t.flatten()
        idx = sample_idx(flat.numel(), 8192, 7)
        out[name + "_idx"] = idx
        out[name + "_val"] = flat[idx].numpy()
        out[name + "_stats"] = np.array([t.mean().item(), t.std().item(), t.min().item(), t.max().item()], np.float32)
    np.savez_compressed(os.path.join(OUT, "full_vits.npz"), **out)
    print("full_vits done", out["depth_m1_stats"])
    cfg4k()
    variants()
    vitl()
    vitl((4, 4), 8, "cfg2_vitl_ref.npz")
    vitl((8, 8), 8, "cfg3_vitl_ref.npz")


if __name__ == "__main__":
    main()
