"""Wiring + weight-packing tests of the engine on CPU: patchfusion_amd.engine driven by the torch
reference op set (tests/fake_ops.py) must reproduce the oracle / the reference golden fixtures.
No HIP kernel runs here (those are covered by the -m gpu tests); this is NOT a product path."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import pf_oracle
from patchfusion_amd.config import make_config
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
from tests.fake_ops import ops as fake_ops

TINY = ("vits", (112, 154), (448, 616), (2, 2))


@pytest.fixture(scope="module")
def tiny():
    cfg = make_config(*TINY)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops).eval()
    print(m.load_state_dict(sd, strict=True))
    img = torch.rand(1, 3, *TINY[2], generator=torch.Generator().manual_seed(1234))
    return cfg, sd, m, img


def nchw(t):
    return t.permute(0, 3, 1, 2) if t.dim() == 4 else t


def test_state_dict_keys_match_spec(tiny):
    cfg, sd, m, _ = tiny
    assert list(m.state_dict().keys()) == list(sd.keys())
    assert set(m.get_save_dict()) == {k for k in sd if "coarse_branch" not in k and "fine_branch" not in k}


def test_branch_stages_match_oracle(tiny):
    cfg, sd, m, img = tiny
    lr = m.resizer(img)
    ot, et = {}, {}
    od, of = pf_oracle.branch_forward(sd, "coarse_branch.", lr, cfg["coarse_branch"], ot)
    st = m._coarse(lr, et)
    for k in ("vit_tokens_in", "vit_block0", "vit_block11"):
        assert (ot[k] - et[k]).abs().max() < 2e-4, k
    for i in range(4):
        a = ot[f"vit_out{i}"]
        b = et[f"vit_out{i}"].reshape(a.shape)
        assert (a - b).abs().max() < 2e-4
    for i, (a, b) in enumerate(zip(of, st["feats"])):
        assert (a - nchw(b)).abs().max() < 5e-4, i
    for i in range(4):
        assert (ot[f"bins_centers{i}"] - nchw(et[f"bins_centers{i}"])).abs().max() < 1e-4
    assert (od - st["depth"]).abs().max() < 1e-4
    g = pf_oracle.g2l_all(sd, of)
    for i, (a, b) in enumerate(zip(g, st["g2l"])):
        assert (a - nchw(b)).abs().max() < 5e-4, i


@pytest.mark.parametrize("mode", ["m1", "m2", "r4"])
def test_end_to_end_matches_golden(tiny, golden_dir, mode):
    cfg, sd, m, img = tiny
    g = np.load(os.path.join(golden_dir, "tiny_vits.npz"))
    lr = m.resizer(img)
    random.seed(5621)
    d, aux = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=mode, process_num=2)
    ref = g[f"depth_{mode}"]
    assert tuple(d.shape) == (1, 1) + ref.shape
    err = np.abs(d[0, 0].numpy() - ref).max()
    assert err < 2e-5, err
    assert aux["depth_pred"] is d and aux["rgb"] is lr


@pytest.mark.skipif(os.environ.get("PF_TEST_FAST") == "1", reason="about one CPU-minute; PF_TEST_FAST=1 skips it")
@pytest.mark.parametrize("name,split,mode", [("c0_2x2_r4", (2, 2), "r4")] +
                         ([("c1_4x4_m1", (4, 4), "m1")] if os.environ.get("PF_TEST_FULL") == "1" else []))   # the 4x4 case: GPU suite + oracle test
def test_host_logic_at_baseline_configs_0_and_1(golden_dir, name, split, mode):
    """BASELINE.json configs[0] / [1] at 2160x3840 through the product's host code (tiling, tile tables, random-tile
    schedule, stitcher, weight packing) with the torch stand-in ops: 16384 sampled outputs of the reference's own runs."""
    g = np.load(os.path.join(golden_dir, "cfg4k_vits.npz"))
    cfg = make_config("vits", (392, 518), (2160, 3840), split)
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops).eval()
    m.load_state_dict(synthetic_state_dict(patchfusion_spec(cfg), 0), strict=True)
    img = torch.rand(1, 3, 2160, 3840, generator=torch.Generator().manual_seed(1234))
    random.seed(5621)
    with torch.no_grad():
        d, _ = m(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode=mode, process_num=4)
    assert tuple(d.shape[2:]) == tuple(int(v) for v in g[name + "_shape"])
    v = d.flatten()[torch.from_numpy(g[name + "_idx"])].numpy()
    assert np.abs(v - g[name + "_val"]).max() <= 1e-5


def test_schedule_properties_tiny(tiny):
    """The same property checker the GPU runs at the headline size (tests/schedule_props.py)."""
    from tests import schedule_props
    cfg, sd, m, img = tiny
    schedule_props.check(m, m.resizer(img), img, cfg, process_num=4, max_tol=1e-5, mean_tol=1e-6)


def test_errors_match_reference_contract(tiny):
    cfg, sd, m, img = tiny
    with pytest.raises(AssertionError):
        m.prepare_tile_cfg((450, 616), (2, 2))
    with pytest.raises(AssertionError):
        m(mode="infer", image_lr=m.resizer(img), image_hr=torch.cat([img, img]), cai_mode="m1")
    bad = make_config(*TINY)
    bad["coarse_branch"]["type"] = "Nope"
    with pytest.raises(NotImplementedError):
        PatchFusion(bad, ops=fake_ops)
    bad = make_config(*TINY)
    bad["coarse_branch"]["bin_centers_type"] = "xyz"
    with pytest.raises(ValueError):
        PatchFusion(bad, ops=fake_ops)


def test_product_path_has_no_cpu_fallback():
    cfg = make_config(*TINY)
    m = PatchFusion(cfg)   # default ops = hip_ops
    with pytest.raises(Exception):
        m(mode="infer", image_lr=torch.zeros(1, 3, 112, 154), image_hr=torch.zeros(1, 3, 448, 616))


@pytest.mark.parametrize("kind,atype,akind", [("normed", "inv", "mean"), ("hybrid1", "exp", "sum"), ("hybrid2", "inv", "sum"), ("softplus", "exp", "mean")])
def test_bin_center_variants_match_reference_golden(golden_dir, kind, atype, akind):
    """bin_centers_type / attractor_type / attractor_kind wiring (engine.BinsHead) against the reference-made fixture"""
    from oracle.make_golden import variant_case
    g = np.load(os.path.join(golden_dir, "variants_vits.npz"))
    cfg, sd, img = variant_case(kind, atype, akind)
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops).eval()
    m.load_state_dict(sd, strict=True)
    d, _ = m(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode="m1", process_num=2)
    ref = g[f"{kind}_depth_m1"]
    assert tuple(d.shape[2:]) == ref.shape
    assert np.abs(d[0, 0].numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), kind
    ref = g[f"{kind}_coarse_depth"]
    assert np.abs(m._coarse_state["depth"][0].numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_bin_centers_type_error_contract():
    cfg = make_config(*TINY)
    cfg["coarse_branch"]["bin_centers_type"] = "bogus"
    with pytest.raises(ValueError, match="bin_centers_type should be one of"):
        PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops)
