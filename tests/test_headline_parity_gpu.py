"""pytest -m gpu: parity AT THE CONFIGURATION THE BENCH TIMES -- BASELINE.json configs[2]: Depth-Anything ViT-L,
2160x3840 image, 4x4 regular tiles (cai_mode m1), process_num 8 -- in both compute modes, against the oracle
(oracle/pf_oracle.py, the restatement pinned to the reference's own Python) evaluated with torch on the same GPU.

The oracle needs ~2 minutes for all 16 tiles on the GPU, so a SAMPLE of the output is checked: the whole coarse pass
(coarse depth = every pixel of the low-resolution prediction) and the four diagonal tiles (0, 5, 10, 15) of the 4x4
grid = 4 x 392 x 518 pixels of the final stitched map (in m1 every pixel of the map belongs to exactly one tile, so
map == tile depth up to the 1-ulp `d*m/m` of the running average).

Stated tolerances, in DEPTH UNITS (the synthetic-weight model predicts depths in [0.67, 0.79], std 0.012):
    f32  (exact mode, the headline precision; also the `f32_mfma_only` dispatch PF_LINEAR_SPLIT3=0 PF_WINO_SPLIT3=0 that bench.py publishes):
         max |delta| <= 2e-5, p99 <= 1e-5, mean <= 2e-6         measured 1.1e-6 / 5.4e-7 / 1.7e-7 (round 3, all 16 tiles) -- ~20x the measurement
    bf16 (fast mode, secondary bench figure):    max <= 5e-3, p99 <= 2e-3, mean <= 6e-4     measured 2.5e-3 / 1.1e-3 / 3.1e-4
Round 4 adds, at the same configuration: FEATURE-level parity of two tiles (the six fine-branch pyramid maps and the six fused maps of the
guided-fusion U-Net, relative rms <= 1e-5 against the oracle) and one whole-net run on a LARGE-DYNAMIC-RANGE weight set
(tests/dynamic_range.py: per-channel scales over 1e-3 ... 1e3 between paired layers, same function in exact arithmetic).
The bf16 budget is 2x the measured error; profiles/r2_precision_probe.json holds the per-stage growth (features carry ~1 %
relative rms error after 24 bf16 ViT blocks, no stage amplifies; the f32 metric-bins head maps it to 5e-4 relative depth).
The measured numbers of each run are written to gpurun_out/r6_headline_parity.json.

PF_HEADLINE_ALL=1 checks ALL 16 tiles live (two extra minutes of oracle time on the GPU; run once per round by the builder, result in
profiles/r6_headline_parity.json) and writes tests/golden-format samples of the oracle's 16 tiles to gpurun_out/headline_vitl_sampled.npz;
the committed copy (tests/golden/headline_vitl_sampled.npz: 4096 pixels of EVERY tile + 8192 of the coarse depth) is what
test_configs2_all_16_tiles_match_sampled_oracle_fixture checks in every run without re-running the oracle.
"""
import json
import os

import pytest
import torch

from oracle import pf_oracle
from patchfusion_amd.config import make_config
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = tuple(range(16)) if os.environ.get("PF_HEADLINE_ALL", "0") == "1" else (0, 5, 10, 15)
SAMPLES = 4096
TOL = {"fp32": dict(max=2e-5, p99=1e-5, mean=2e-6), "bf16": dict(max=5e-3, p99=2e-3, mean=6e-4)}
TOL["fp32_mfma_only"] = TOL["fp32"]
F32_MFMA_ONLY_ENV = {"PF_LINEAR_SPLIT3": "0", "PF_WINO_SPLIT3": "0"}          # every GEMM on the f32 MFMA (bench.py `f32_mfma_only`)
FEATURE_REL_RMS = 1e-5
# the coarse branch's own depth is an INTERMEDIATE (it enters the fusion net as one of 5 input channels); with the synthetic
# weights its bin softmax is far more selective than the fusion head's (depths 0.56..0.99, std 0.039), so isolated pixels
# near a tie between bins move by up to 0.05 in bf16 (measured max 0.053, p99 7.7e-3, mean 1.3e-3); budget = 2x measured
# near a tie between bins move by up to 0.05 in bf16 (measured max 0.053, p99 7.7e-3, mean 1.3e-3); budget = 2x measured.  The same
# sensitivity shows in f32: two f32 evaluations that differ only in summation order (engine vs torch/MIOpen on the GPU) agree
# to 1.6e-4 max / 3.0e-5 p99 / 5.1e-6 mean on this map while the final map agrees to 1.0e-5.
# (round 4: 9.8e-6 / 2.3e-6 / 4.3e-7 measured in f32 with every GEMM in split precision or on the f32 MFMA; budget ~10x)
TOL_COARSE = {"fp32": dict(max=1e-4, p99=2e-5, mean=4e-6), "bf16": dict(max=0.11, p99=1.6e-2, mean=2.6e-3)}
TOL_COARSE["fp32_mfma_only"] = TOL_COARSE["fp32"]


def _oracle_eval(sd, img, tiles, with_features=0):
    """oracle (torch on the GPU) at configs[2]: coarse depth, the final depth of `tiles`, and -- for the first `with_features` tiles -- the
    resized crops, the six fine-branch feature maps and the six fused maps (NCHW float32)"""
    cfg = make_config("vitl", (392, 518), (2160, 3840), (4, 4))
    sdg = {k: v.cuda() for k, v in sd.items()}
    orc = pf_oracle.Oracle(cfg, sdg)
    extra = {}
    with torch.no_grad():
        lr = orc.resizer(img)
        orc.coarse_depth, orc.coarse_feats = pf_oracle.branch_forward(sdg, "coarse_branch.", lr, cfg["coarse_branch"])
        orc.g2l = pf_oracle.g2l_all(sdg, orc.coarse_feats)
        tile_cfg = pf_oracle.prepare_tile_cfg(orc.ps, cfg["image_raw_shape"], cfg["patch_split_num"])
        hr, wr = tile_cfg["patch_raw_shape"]
        crops, boxes = [], []
        for t in tiles:
            h, w = (t // 4) * hr, (t % 4) * wr
            crops.append(orc.resizer(img[:, :, h:h + hr, w:w + wr])[0])
            boxes.append([w, h, w + wr, h + hr])
        crops = torch.stack(crops)
        bt = torch.tensor(boxes, device="cuda").int()
        ref_tiles = orc._predict(crops, bt, tile_cfg, 2)[:, 0].clone()
        ref_coarse = orc.coarse_depth.clone()
        if with_features:
            n = with_features
            _, ffeats = pf_oracle.any_branch_forward(sdg, "fine_branch.", crops[:n], cfg["fine_branch"], None)
            orc.taps = {}
            orc._predict(crops[:n], bt[:n], tile_cfg, n)
            extra = dict(crops=crops[:n].clone(), boxes=boxes[:n], fine_feats=[f.float().clone() for f in ffeats],
                         fused=[orc.taps[f"gf_out{i}"].float().clone() for i in range(6)])
            orc.taps = None
    del orc, sdg
    torch.cuda.empty_cache()
    return cfg, ref_coarse, ref_tiles, extra


@pytest.fixture(scope="module")
def oracle_sample():
    cfg = make_config("vitl", (392, 518), (2160, 3840), (4, 4))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).cuda()
    cfg, ref_coarse, ref_tiles, extra = _oracle_eval(sd, img, TILES, with_features=2)
    if len(TILES) == 16:        # sampled fixture of the oracle's whole map (see the module docstring)
        import numpy as np
        g = torch.Generator().manual_seed(99)
        idx = torch.stack([torch.randperm(392 * 518, generator=g)[:SAMPLES] for _ in range(16)])
        cidx = torch.randperm(ref_coarse.numel(), generator=g)[:8192]
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "headline_vitl_sampled.npz"), tile_idx=idx.numpy().astype(np.int32),
                                tile_val=torch.gather(ref_tiles.flatten(1).cpu(), 1, idx).numpy(), coarse_idx=cidx.numpy().astype(np.int32),
                                coarse_val=ref_coarse.flatten().cpu()[cidx].numpy())
        except OSError:
            pass
    return cfg, sd, img, ref_coarse, ref_tiles, extra


class _env:
    def __init__(self, kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _record(key, rec):
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "r6_headline_parity.json")
        allrec = json.load(open(path)) if os.path.exists(path) else {}
        allrec[key] = rec
        json.dump(allrec, open(path, "w"), indent=1)
    except OSError:
        pass


def _stats(diff):
    d = diff.flatten().float()
    return dict(max=float(d.max()), p99=float(torch.quantile(d[::3], 0.99)), mean=float(d.mean()))


@pytest.mark.parametrize("variant", ["fp32", "fp32_mfma_only", "bf16"])
def test_configs2_vitl_4k_p16_matches_oracle(oracle_sample, variant):
    cfg, sd, img, ref_coarse, ref_tiles, _ = oracle_sample
    dtype = "bf16" if variant == "bf16" else "fp32"
    with _env(F32_MFMA_ONLY_ENV if variant == "fp32_mfma_only" else {}):
        m = PatchFusion(cfg, compute_dtype=dtype).eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        lr = m.resizer(img)
        with torch.no_grad():
            d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=8)
        torch.cuda.synchronize()
    assert tuple(d.shape) == (1, 1, 4 * 392, 4 * 518)
    got = torch.stack([d[0, 0, (t // 4) * 392:(t // 4 + 1) * 392, (t % 4) * 518:(t % 4 + 1) * 518] for t in TILES])
    st_tiles = _stats((got - ref_tiles).abs())
    st_coarse = _stats((m._coarse_state["depth"] - ref_coarse).abs())
    _record(variant, dict(config="BASELINE configs[2]: DA-vitl 2160x3840 4x4 m1 process_num=8", dtype=dtype, variant=variant, tiles=list(TILES),
                          final_map_vs_oracle=st_tiles, coarse_depth_vs_oracle=st_coarse, tolerance=TOL[variant], tolerance_coarse=TOL_COARSE[variant],
                          ref_depth_range=[float(ref_tiles.min()), float(ref_tiles.max())], ref_depth_std=float(ref_tiles.std())))
    for name, st, tol in (("final map", st_tiles, TOL[variant]), ("coarse depth", st_coarse, TOL_COARSE[variant])):
        assert st["max"] <= tol["max"] and st["p99"] <= tol["p99"] and st["mean"] <= tol["mean"], (variant, name, st, tol)
    del m
    torch.cuda.empty_cache()


def _rel_rms(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def test_configs2_vitl_feature_level_parity_two_tiles(oracle_sample):
    """ViT-L, 392 x 518, tiles 0 and 5 of configs[2]: the six maps of the fine-branch pyramid (x_d0, r4, r3, r2, r1, out_conv -- what
    PatchFusion.fine_forward returns, patchfusion.py:208-218) and the six fused maps of the guided-fusion U-Net (guided_fusion_model.py:198-203)
    against the oracle, relative rms <= 1e-5 each.  The final depth is two scalars per pixel behind a softmax head; this pins the 24 ViT-L
    blocks, the DPT head and every fusion level directly."""
    cfg, sd, img, _, _, extra = oracle_sample
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    tile_cfg = m.prepare_tile_cfg(cfg["image_raw_shape"], cfg["patch_split_num"])
    with torch.no_grad():
        m.coarse_forward(m.resizer(img))
        crops = extra["crops"]
        _, ffeats = m.fine_forward(crops)
        taps = {}
        m.infer_forward(crops, m._rois(extra["boxes"], tile_cfg, crops.device), taps=taps)
    torch.cuda.synchronize()
    rec = {}
    for i, (a, b) in enumerate(zip(ffeats, extra["fine_feats"])):
        assert tuple(a.shape) == tuple(b.shape), (i, a.shape, b.shape)
        rec[f"fine_L{i}"] = _rel_rms(a, b)
    for i, b in enumerate(extra["fused"]):
        a = taps[f"gf_out{i}"].float().permute(0, 3, 1, 2)
        assert tuple(a.shape) == tuple(b.shape), (i, a.shape, b.shape)
        rec[f"fused_{i}"] = _rel_rms(a, b)
    _record("feature_level_rel_rms", rec)
    print("MEASURED feature-level relative rms:", {k: f"{v:.2e}" for k, v in rec.items()})
    assert max(rec.values()) <= FEATURE_REL_RMS, rec
    del m
    torch.cuda.empty_cache()


def test_configs2_large_dynamic_range_weights_match_oracle(oracle_sample):
    """the whole net at configs[2] on a weight set whose paired layers exchange per-channel scales over 1e-3 ... 1e3 (tests/dynamic_range.py):
    q / k / v planes, attention outputs, ResidualConvUnit and double-conv intermediates span six decades across channels.  Engine (default f32
    dispatch: split-precision GEMMs / attention, Winograd layers) against the oracle on the SAME weights: coarse depth and tiles 0, 5 of the
    final map within the f32 budget."""
    from tests.dynamic_range import widen_dynamic_range
    cfg, sd, img, _, _, _ = oracle_sample
    sw = widen_dynamic_range(sd, seed=7)
    _, ref_coarse, ref_tiles, _ = _oracle_eval(sw, img, (0, 5))
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sw, strict=True)
    m = m.cuda()
    with torch.no_grad():
        d, _ = m(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode="m1", process_num=8)
    torch.cuda.synchronize()
    got = torch.stack([d[0, 0, (t // 4) * 392:(t // 4 + 1) * 392, (t % 4) * 518:(t % 4 + 1) * 518] for t in (0, 5)])
    st_tiles = _stats((got - ref_tiles).abs())
    st_coarse = _stats((m._coarse_state["depth"] - ref_coarse).abs())
    _record("large_dynamic_range", dict(final_map_vs_oracle=st_tiles, coarse_depth_vs_oracle=st_coarse, tolerance=TOL["fp32"],
                                        scales="per-channel 1e-3 ... 1e3 between paired layers (tests/dynamic_range.py, seed 7)"))
    print("MEASURED large dynamic range:", st_tiles, st_coarse)
    for name, st, tol in (("final map", st_tiles, TOL["fp32"]), ("coarse depth", st_coarse, TOL_COARSE["fp32"])):
        assert st["max"] <= tol["max"] and st["p99"] <= tol["p99"] and st["mean"] <= tol["mean"], (name, st, tol)
    del m
    torch.cuda.empty_cache()


def test_configs2_all_16_tiles_match_sampled_oracle_fixture(golden_dir):
    """EVERY tile of the configs[2] map (and the coarse depth) against the committed samples of the oracle's output
    (tests/golden/headline_vitl_sampled.npz, written by the PF_HEADLINE_ALL=1 run of this module on a GPU box): the f32 bound of the
    four-tile live check, for all 16 tiles, without the two minutes of oracle time."""
    import numpy as np
    path = os.path.join(golden_dir, "headline_vitl_sampled.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/headline_vitl_sampled.npz not generated yet")
    g = np.load(path)
    cfg = make_config("vitl", (392, 518), (2160, 3840), (4, 4))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).cuda()
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    with torch.no_grad():
        d, _ = m(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode="m1", process_num=8)
    tiles = torch.stack([d[0, 0, (t // 4) * 392:(t // 4 + 1) * 392, (t % 4) * 518:(t % 4 + 1) * 518] for t in range(16)]).flatten(1).cpu()
    got = torch.gather(tiles, 1, torch.from_numpy(g["tile_idx"]).long()).numpy()
    err = np.abs(got - g["tile_val"])
    cerr = np.abs(m._coarse_state["depth"].flatten().cpu().numpy()[g["coarse_idx"]] - g["coarse_val"])
    print(f"MEASURED all 16 tiles vs sampled oracle: max {err.max():.3e} mean {err.mean():.3e}; per-tile max {err.max(axis=1).round(7).tolist()}; coarse max {cerr.max():.3e}")
    assert err.max() <= TOL["fp32"]["max"] and err.mean() <= TOL["fp32"]["mean"], (err.max(), err.mean())
    assert cerr.max() <= TOL_COARSE["fp32"]["max"], cerr.max()
    del m
    torch.cuda.empty_cache()


def test_vitl_2x2_matches_reference_made_fixture(golden_dir):
    """The HIP path against the REFERENCE ITSELF at the headline architecture (round-4 review, missing #5): DA-vitl, 392 x 518, one 2160 x 3840 image
    in 2 x 2 tiles, m1, process_num 4 -- tests/golden/headline_vitl_ref.npz holds samples of what the reference's own PatchFusion.forward(mode='infer')
    (patchfusion.py:401-453) returned for the seeded weights / image (oracle/make_golden.py vitl; the same file pins the oracle on the CPU,
    tests/test_oracle_golden.py).  Final map and coarse depth inside the f32 budget of the headline test, two coarse feature levels at 1e-5 relative rms."""
    import numpy as np
    path = os.path.join(golden_dir, "headline_vitl_ref.npz")
    g = np.load(path)
    cfg = make_config("vitl", (392, 518), (2160, 3840), (2, 2))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).cuda()
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    with torch.no_grad():
        lr = m.resizer(img)
        cd, cf = m.coarse_forward(lr)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=4)
    torch.cuda.synchronize()
    rec = {}
    for name, t in (("depth_m1", d), ("coarse_depth", cd), ("coarse_feat3", cf[3]), ("coarse_feat5", cf[5])):
        assert tuple(t.shape) == tuple(int(v) for v in g[name + "_shape"]), (name, t.shape)
        got = t.flatten().cpu()[torch.from_numpy(g[name + "_idx"]).long()].double().numpy()
        ref = g[name + "_val"].astype(np.float64)
        rec[name] = dict(max_abs=float(np.abs(got - ref).max()), mean_abs=float(np.abs(got - ref).mean()),
                         rel_rms=float(np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30)))
    _record("vitl_2x2_vs_reference_fixture", rec)
    print("MEASURED ViT-L 2x2 vs the reference-made fixture:", {k: {a: f"{b:.2e}" for a, b in v.items()} for k, v in rec.items()})
    assert rec["depth_m1"]["max_abs"] <= TOL["fp32"]["max"] and rec["depth_m1"]["mean_abs"] <= TOL["fp32"]["mean"], rec
    assert rec["coarse_depth"]["max_abs"] <= TOL_COARSE["fp32"]["max"], rec
    assert rec["coarse_feat3"]["rel_rms"] <= FEATURE_REL_RMS and rec["coarse_feat5"]["rel_rms"] <= FEATURE_REL_RMS, rec
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fname,split", [("cfg2_vitl_ref.npz", (4, 4)), ("cfg3_vitl_ref.npz", (8, 8))])
def test_benched_configs_match_reference_made_fixtures(golden_dir, fname, split):
    """The HIP path against the REFERENCE ITSELF at the benched configurations (round-5 review, missing #2): BASELINE.json configs[2] (DA-vitl, 2160 x 3840,
    4 x 4 tiles, m1, process_num 8 -- what bench.py times) and the configs[3] geometry (8 x 8 tiles, 64 patches; one GPU runs all of them here, the 8-GPU
    job shards them, tests/test_dist_cpu.py) -- tests/golden/cfg2_vitl_ref.npz / cfg3_vitl_ref.npz hold samples of what the reference's own
    PatchFusion.forward(mode='infer') (patchfusion.py:401-453) returned for the seeded weights / image (oracle/make_golden.py cfg2 / cfg3; the same files
    pin the oracle on the CPU, tests/test_oracle_golden.py).  Every sample of the final map and of the coarse depth inside the f32 budget of the headline
    test, two coarse feature levels at 1e-5 relative rms."""
    import numpy as np
    g = np.load(os.path.join(golden_dir, fname))
    cfg = make_config("vitl", (392, 518), (2160, 3840), split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234)).cuda()
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    with torch.no_grad():
        lr = m.resizer(img)
        cd, cf = m.coarse_forward(lr)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=8)
    torch.cuda.synchronize()
    rec = {}
    for name, t in (("depth_m1", d), ("coarse_depth", cd), ("coarse_feat3", cf[3]), ("coarse_feat5", cf[5])):
        assert tuple(t.shape) == tuple(int(v) for v in g[name + "_shape"]), (name, t.shape)
        got = t.flatten().cpu()[torch.from_numpy(g[name + "_idx"]).long()].double().numpy()
        ref = g[name + "_val"].astype(np.float64)
        rec[name] = dict(max_abs=float(np.abs(got - ref).max()), mean_abs=float(np.abs(got - ref).mean()),
                         rel_rms=float(np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30)))
    _record(f"{fname[:4]}_vs_reference_fixture", rec)
    print(f"MEASURED ViT-L {split[0]}x{split[1]} (process_num 8) vs the reference-made fixture:", {k: {a: f"{b:.2e}" for a, b in v.items()} for k, v in rec.items()})
    assert rec["depth_m1"]["max_abs"] <= TOL["fp32"]["max"] and rec["depth_m1"]["mean_abs"] <= TOL["fp32"]["mean"], rec
    assert rec["coarse_depth"]["max_abs"] <= TOL_COARSE["fp32"]["max"], rec
    assert rec["coarse_feat3"]["rel_rms"] <= FEATURE_REL_RMS and rec["coarse_feat5"]["rel_rms"] <= FEATURE_REL_RMS, rec
    del m
    torch.cuda.empty_cache()
