"""pytest -m gpu: the whole PatchFusion hot path on the MI355X HIP engine (through the C ABI) against
 (a) the committed golden fixtures generated from the REFERENCE's own Python (tests/golden/*.npz),
 (b) the oracle restatement (oracle/pf_oracle.py) on the same seeded inputs.
Stated tolerances on the final metric depth, in DEPTH UNITS (synthetic-weight nets: depths 0.4..1.0, std 0.011..0.014):
   fp32 mode: max |d - ref| <= F32_TOL = 5e-5 (float32-grade arithmetic everywhere; only summation order differs; measured ~1e-6 .. 1e-5 --
              round 3 asserted 2e-4 here, 20-200x the measurement: a 50x numerical regression would have passed)
   bf16 mode: max |d - ref| <= 5e-3 and mean |d - ref| <= 6e-4 = 2x the error measured in round 2 (tiny: max 1.7e-3 / mean
              2.6e-4; ViT-L tiles: max 2.4e-3 / mean 3.1e-4; the configuration the bench times: tests/test_headline_parity_gpu.py)
"""
BF16_MAX, BF16_MEAN = 5e-3, 6e-4
F32_TOL = 5e-5
import os
import random

import numpy as np
import pytest
import torch

from oracle import pf_oracle
from patchfusion_amd.config import make_config
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict

pytestmark = pytest.mark.gpu
TINY = ("vits", (112, 154), (448, 616), (2, 2))


def build(enc, ps, raw, split, dtype):
    cfg = make_config(enc, ps, raw, split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    m = PatchFusion(cfg, compute_dtype=dtype).eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234))
    return cfg, sd, m, img


@pytest.mark.parametrize("mode", ["m1", "m2", "r4"])
def test_tiny_fp32_matches_reference_golden(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "tiny_vits.npz"))
    cfg, sd, m, img = build(*TINY, "fp32")
    lr = m.resizer(img).cuda()
    random.seed(5621)
    d, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode=mode, process_num=2)
    ref = g[f"depth_{mode}"]
    assert tuple(d.shape[2:]) == ref.shape
    err = np.abs(d[0, 0].cpu().numpy() - ref).max()
    assert err <= F32_TOL, f"{mode}: {err}"


def test_vitb_tiny_fp32_vs_oracle():
    """Depth-Anything ViT-B (D=768, 12 heads, C=128): the middle encoder of the reference's three configs
    (configs/patchfusion_depthanything/depthanything_vitb_patchfusion_u4k.py), whole path against the oracle."""
    cfg, sd, m, img = build("vitb", (112, 154), (448, 616), (2, 2), "fp32")
    lr = m.resizer(img)
    d, _ = m(mode="infer", image_lr=lr.cuda(), image_hr=img.cuda(), cai_mode="m1", process_num=4)
    ref = pf_oracle.Oracle(cfg, sd).infer(lr, img, "m1", 4)
    assert float((d.cpu() - ref).abs().max()) < F32_TOL


def test_tiny_fp32_stage_parity_vs_oracle():
    cfg, sd, m, img = build(*TINY, "fp32")
    lr = m.resizer(img)
    ot, et = {}, {}
    od, of = pf_oracle.branch_forward(sd, "coarse_branch.", lr, cfg["coarse_branch"], ot)
    st = m._coarse(lr.cuda(), et)
    nchw = lambda t: t.float().cpu().permute(0, 3, 1, 2)
    for k in ("vit_tokens_in", "vit_block0", "vit_block11"):
        assert (ot[k] - et[k].float().cpu()).abs().max() < 1e-3, k
    for i, (a, b) in enumerate(zip(of, st["feats"])):
        assert (a - nchw(b)).abs().max() < 2e-3, i
    assert (od - st["depth"].cpu()).abs().max() < F32_TOL
    g2l = pf_oracle.g2l_all(sd, of)
    for i, (a, b) in enumerate(zip(g2l, st["g2l"])):
        assert (a - nchw(b)).abs().max() < 2e-3, i


def test_tiny_bf16_within_stated_tolerance(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_vits.npz"))
    cfg, sd, m, img = build(*TINY, "bf16")
    lr = m.resizer(img).cuda()
    d, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode="m1", process_num=4)
    ref = g["depth_m1"]
    diff = np.abs(d[0, 0].float().cpu().numpy() - ref)
    print(f"MEASURED tiny bf16 vs reference golden: max {diff.max():.3e} p99 {np.quantile(diff, 0.99):.3e} mean {diff.mean():.3e} std(ref) {ref.std():.3e}")
    assert diff.max() <= BF16_MAX and diff.mean() <= BF16_MEAN, (diff.max(), diff.mean(), ref.std())


def test_full_size_vits_fp32_matches_reference_golden(golden_dir):
    """392x518 process shape (the shipped configs), 784x1036 image, 2x2 tiles: 8192 sampled outputs of
    the reference + the coarse depth."""
    g = np.load(os.path.join(golden_dir, "full_vits.npz"))
    cfg, sd, m, img = build("vits", (392, 518), (784, 1036), (2, 2), "fp32")
    lr = m.resizer(img).cuda()
    random.seed(5621)
    d, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode="m1", process_num=4)
    v = d.flatten().cpu()[torch.from_numpy(g["depth_m1_idx"])].numpy()
    assert np.abs(v - g["depth_m1_val"]).max() <= F32_TOL
    c = m._coarse_state["depth"].flatten().cpu()[torch.from_numpy(g["coarse_depth_idx"])].numpy()
    assert np.abs(c - g["coarse_depth_val"]).max() <= F32_TOL


@pytest.mark.parametrize("name,split,mode", [("c0_2x2_r4", (2, 2), "r4"), ("c1_4x4_m1", (4, 4), "m1")])
def test_baseline_configs_0_and_1_match_reference_golden(golden_dir, name, split, mode):
    """BASELINE.json configs[0] (DA-vits, one 2160x3840 image, 2x2 tiles + random tiles `r4`, 13 patches) and configs[1]
    (DA-vits, 4K, 4x4 regular tiling, batch 4) at their real size: 16384 sampled outputs of the REFERENCE's own run
    (tests/golden/cfg4k_vits.npz, oracle/make_golden.py cfg4k), fp32 mode, same 2e-4 bound as the small fixtures."""
    g = np.load(os.path.join(golden_dir, "cfg4k_vits.npz"))
    cfg, sd, m, img = build("vits", (392, 518), (2160, 3840), split, "fp32")
    lr = m.resizer(img).cuda()
    random.seed(5621)
    d, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode=mode, process_num=4)
    assert tuple(d.shape[2:]) == tuple(int(v) for v in g[name + "_shape"])
    v = d.flatten().cpu()[torch.from_numpy(g[name + "_idx"])].numpy()
    assert np.abs(v - g[name + "_val"]).max() <= F32_TOL
    del m
    torch.cuda.empty_cache()


def test_vitl_patch_batch_vs_oracle_on_gpu():
    """Depth-Anything ViT-L (the headline model) at the full 392x518 process shape, one fine+fusion batch
    of 2 tiles, engine (fp32 and bf16) vs the oracle evaluated with torch on the same GPU."""
    cfg = make_config("vitl", (392, 518), (784, 1036), (2, 2))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, 784, 1036, generator=torch.Generator().manual_seed(1234)).cuda()
    sdg = {k: v.cuda() for k, v in sd.items()}
    orc = pf_oracle.Oracle(cfg, sdg)
    lr = orc.resizer(img)
    ref = orc.infer(lr, img, "m1", 2)[0, 0]
    for dtype, tol_max, tol_mean in (("fp32", 5e-4, 5e-5), ("bf16", None, None)):
        m = PatchFusion(cfg, compute_dtype=dtype).eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=2)
        diff = (d[0, 0] - ref).abs()
        s = float(ref.std())
        print(f"MEASURED vitl 2-tile {dtype} vs oracle: max {float(diff.max()):.3e} p99 {float(torch.quantile(diff.flatten()[::3], 0.99)):.3e} "
              f"mean {float(diff.mean()):.3e} std(ref) {s:.3e}")
        if dtype == "fp32":
            assert float(diff.max()) <= tol_max and float(diff.mean()) <= tol_mean, (float(diff.max()), float(diff.mean()))
        else:
            assert float(diff.max()) <= BF16_MAX and float(diff.mean()) <= BF16_MEAN, (float(diff.max()), float(diff.mean()), s)
        del m
        torch.cuda.empty_cache()


def test_headline_size_schedule_properties():
    """BASELINE.json configs[2] itself (Depth-Anything ViT-L, 2160x3840, 4x4 tiles, process_num 8, bf16): the oracle
    needs minutes per tile on a CPU, so the full-size pass is checked through size-independent properties
    (tests/schedule_props.py): determinism, stream-schedule invariance (bit-exact), batch-size invariance within the
    bf16 budget, finite / in-range / reensemble-shaped output.  (The same configuration is compared with the oracle in
    tests/test_headline_parity_gpu.py.)"""
    from tests import schedule_props
    cfg, sd, m, img = build("vitl", (392, 518), (2160, 3840), (4, 4), "bf16")
    img = img.cuda()
    lr = m.resizer(img)
    # batch invariance: two bf16 passes whose layers run on different tile shapes (other summation order) may each carry
    # the bf16 error budget -> allow 1.2x the single-pass budget between them
    schedule_props.check(m, lr, img, cfg, process_num=8, max_tol=1.2 * BF16_MAX, mean_tol=1.2 * BF16_MEAN)
    del m
    torch.cuda.empty_cache()


def test_baseline_pretrain_fine_and_coarse_vs_oracle():
    """BaselinePretrain on the HIP engine: coarse branch and tiled fine branch (m2) vs oracle."""
    from collections import OrderedDict
    from patchfusion_amd.baseline import BaselinePretrain
    from patchfusion_amd.config import zoe_branch_config
    from patchfusion_amd.spec import branch_spec
    ps, raw, split = (112, 154), (448, 616), (2, 2)
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234))
    for target, mode in (("coarse", "m1"), ("fine", "m2")):
        bc = zoe_branch_config("vits", ps)
        spec = OrderedDict()
        branch_spec(spec, f"{target}_branch.", bc)
        sd = synthetic_state_dict(spec, 0)
        m = BaselinePretrain(bc, bc, dict(type="SILogLoss"), 1e-3, 80, raw, ps, split, target=target).eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        lr = m.resizer(img)
        d, _ = m(mode="infer", image_lr=lr.cuda(), image_hr=img.cuda(), cai_mode=mode, process_num=2)
        o = pf_oracle.BaselineOracle(bc, ps, raw, split, sd, target).infer(lr, img, mode, 2)
        assert d.shape == o.shape and float((d.cpu() - o).abs().max()) < F32_TOL, (target, float((d.cpu() - o).abs().max()))


def test_zoe_midas_core_geometry_r_mode_vs_oracle():
    """BASELINE configs[4], encoder-independent part on the HIP engine at the real 384x512 geometry (multiple-of-32 resize,
    GuidedFusion default pyramid 12x16 ... 384x512, C=256) with random tiles (`r4`, 13 patches): ZoeDepth head + fusion net +
    tiling vs the oracle, both fed by the same stand-in relative-depth core (tests/zoe_core_standin.py; the MiDaS/BEiT core
    itself is an un-vendored torch.hub repo -> PARITY UNPINNED)."""
    from patchfusion_amd.config import make_zoe_config
    from tests.zoe_core_standin import StandInCore
    cfg = make_zoe_config((384, 512), (1536, 2048), (2, 2))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    cores = (StandInCore(11), StandInCore(12))
    img = torch.rand(1, 3, 1536, 2048, generator=torch.Generator().manual_seed(1234)).cuda()
    sdg = {k: v.cuda() for k, v in sd.items()}
    for dtype, tol in (("fp32", F32_TOL), ("bf16", 1e-2)):
        m = PatchFusion(cfg, compute_dtype=dtype, core_providers=cores).eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        lr = m.resizer(img)
        assert tuple(lr.shape) == (1, 3, 384, 512)
        random.seed(5621)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="r4", process_num=2)
        random.seed(5621)
        ref = pf_oracle.Oracle(cfg, sdg, core_providers=cores).infer(lr, img, "r4", 2)
        assert d.shape == ref.shape == (1, 1, 1536, 2048)
        err = float((d - ref).abs().max())
        print(f"MEASURED zoe-geometry r4 {dtype} vs oracle: max {err:.3e} mean {float((d - ref).abs().mean()):.3e} std(ref) {float(ref.std()):.3e}")
        assert err <= tol, (dtype, err)
        del m
        torch.cuda.empty_cache()


def test_train_mode_forward_vs_oracle():
    """`forward(mode='train')` (patchfusion.py:372-399) on the HIP engine: forward value of the training step
    (batch of 2 images, one crop each, roi_align with per-sample batch index, fusion, SILogLoss kernel) vs the oracle, which is
    pinned to the reference's own train forward (tests/test_train_forward_cpu.py)."""
    from tests.test_train_forward_cpu import train_batch
    cfg, sd, m, _ = build(*TINY, "fp32")
    image_lr, crops, bboxs, gt = train_batch()
    loss_dict, aux = m(mode="train", image_lr=image_lr.cuda(), image_hr=None, crops_image_hr=crops.cuda(), crop_depths=gt.cuda(),
                       bboxs=bboxs.cuda())
    ref_loss, ref_pred = pf_oracle.Oracle(cfg, sd).train_forward(image_lr, crops, gt, bboxs)
    assert float((aux["depth_pred"].cpu() - ref_pred).abs().max()) < F32_TOL
    assert abs(float(loss_dict["total_loss"]) - float(ref_loss)) < 1e-3 * max(1.0, abs(float(ref_loss)))
    # degenerate mask: <= 1 valid pixel -> 0 (losses.py:38-40)
    z = m.ops.silog_loss(aux["depth_pred"].contiguous(), torch.zeros_like(aux["depth_pred"]), 1e-3, 80)
    assert float(z) == 0.0


@pytest.mark.parametrize("kind,atype,akind", [("normed", "inv", "mean"), ("hybrid1", "exp", "sum"), ("hybrid2", "inv", "sum"), ("softplus", "exp", "mean")])
def test_bin_center_variants_fp32_match_reference_golden(golden_dir, kind, atype, akind):
    """bin_centers_type 'normed' / 'hybrid1' / 'hybrid2' (bounded seed regressor / attractor layer, zoedepth_v1.py:90-104) and
    attractor_type 'exp' / attractor_kind 'sum': whole path on the HIP engine against the reference-made fixture
    (tests/golden/variants_vits.npz).  Depth ranges differ per variant (up to 62), so the 2e-4 bar is relative to max |depth|."""
    from oracle.make_golden import variant_case
    g = np.load(os.path.join(golden_dir, "variants_vits.npz"))
    cfg, sd, img = variant_case(kind, atype, akind)
    m = PatchFusion(cfg, compute_dtype="fp32").eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    d, _ = m(mode="infer", image_lr=m.resizer(img).cuda(), image_hr=img.cuda(), cai_mode="m1", process_num=2)
    ref = g[f"{kind}_depth_m1"]
    err = np.abs(d[0, 0].cpu().numpy() - ref).max()
    assert err <= F32_TOL * max(1.0, np.abs(ref).max()), (kind, err)


def test_baseline_pretrain_train_mode_and_external_core_on_the_engine():
    """BaselinePretrain on the HIP engine: `forward(mode='train')` (baseline_pretrain.py:347-363: branch + SILogLoss kernel, forward
    value) for both targets, and a type-'ZoeDepth' fine branch (external core through the provider hook, multiple-of-32 resizer)
    tiled with m1 -- against BaselineOracle (pinned to the reference's own BaselinePretrain, tests/test_baseline_cpu.py)."""
    from collections import OrderedDict
    from patchfusion_amd.baseline import BaselinePretrain
    from patchfusion_amd.config import zoe_branch_config, zoe_midas_branch_config
    from patchfusion_amd.spec import branch_spec
    from tests.test_baseline_cpu import _train_batch
    from tests.zoe_core_standin import StandInCore
    ps, raw, split = (112, 154), (448, 616), (2, 2)
    for target in ("coarse", "fine"):
        bc = zoe_branch_config("vits", ps)
        spec = OrderedDict()
        branch_spec(spec, f"{target}_branch.", bc)
        sd = synthetic_state_dict(spec, 0)
        m = BaselinePretrain(bc, bc, dict(type="SILogLoss"), 1e-3, 80, raw, ps, split, target=target).eval()
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        x, gt = _train_batch(target)
        kw = dict(depth_gt=gt.cuda()) if target == "coarse" else dict(crops_image_hr=x.cuda(), crop_depths=gt.cuda())
        loss, aux = m(mode="train", image_lr=x.cuda(), image_hr=None, **kw)
        want, pred = pf_oracle.BaselineOracle(bc, ps, raw, split, sd, target).train_forward(x, gt)
        assert float((aux["depth_pred"].cpu() - pred).abs().max()) < F32_TOL
        assert abs(float(loss["total_loss"]) - float(want)) < 1e-3 * max(1.0, abs(float(want))), target
    ps, raw = (96, 128), (384, 512)
    bc = zoe_midas_branch_config(ps)
    spec = OrderedDict()
    branch_spec(spec, "fine_branch.", bc)
    sd = synthetic_state_dict(spec, 0)
    core = StandInCore(21)
    m = BaselinePretrain(bc, bc, None, 1e-3, 80, raw, ps, split, target="fine", core_provider=core).eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(4321))
    lr = m.resizer(img)
    d, _ = m(mode="infer", image_lr=lr.cuda(), image_hr=img.cuda(), cai_mode="m1", process_num=2)
    o = pf_oracle.BaselineOracle(bc, ps, raw, split, sd, "fine", core_provider=core).infer(lr, img, "m1", 2)
    assert d.shape == o.shape and float((d.cpu() - o).abs().max()) < F32_TOL


def test_configs3_geometry_8x8_tiles_vs_oracle_on_gpu():
    """BASELINE.json configs[3] geometry through the HIP kernels: 2160x3840 split 8x8 = 64 tiles whose 270x480 raw crops are
    UP-sampled to 392x518 by crop_resize_planar_kernel, stitched into a 3136x4144 map (DA-vits weights keep the oracle side short).
    Engine: all 64 tiles.  Oracle (torch on this GPU): the coarse pass and six tiles spread over the grid incl. the corners."""
    cfg, sd, m, img = build("vits", (392, 518), (2160, 3840), (8, 8), "fp32")
    img = img.cuda()
    lr = m.resizer(img)
    d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=8)
    assert tuple(d.shape) == (1, 1, 8 * 392, 8 * 518)
    sdg = {k: v.cuda() for k, v in sd.items()}
    orc = pf_oracle.Oracle(cfg, sdg)
    tiles = (0, 7, 27, 36, 56, 63)
    with torch.no_grad():
        # (torch's GPU interpolate and the HIP resize kernel round the 4K -> 392x518 source coordinates differently, ~1e-4 on the image:
        # the oracle's coarse pass gets the engine's image_lr so that this test isolates the tile geometry)
        orc.coarse_depth, orc.coarse_feats = pf_oracle.branch_forward(sdg, "coarse_branch.", lr, cfg["coarse_branch"])
        orc.g2l = pf_oracle.g2l_all(sdg, orc.coarse_feats)
        tile_cfg = pf_oracle.prepare_tile_cfg(orc.ps, cfg["image_raw_shape"], cfg["patch_split_num"])
        hr, wr = tile_cfg["patch_raw_shape"]
        assert (hr, wr) == (270, 480)
        crops, boxes = [], []
        for t in tiles:
            h, w = (t // 8) * hr, (t % 8) * wr
            crops.append(orc.resizer(img[:, :, h:h + hr, w:w + wr])[0])
            boxes.append([w, h, w + wr, h + hr])
        ref = orc._predict(torch.stack(crops), torch.tensor(boxes, device="cuda").int(), tile_cfg, 3)[:, 0]
    got = torch.stack([d[0, 0, (t // 8) * 392:(t // 8 + 1) * 392, (t % 8) * 518:(t % 8 + 1) * 518] for t in tiles])
    err = float((got - ref).abs().max())
    print(f"MEASURED 8x8 geometry (configs[3]), 6 of 64 tiles vs oracle: max {err:.3e}")
    assert err <= F32_TOL, err
    del m, orc, sdg
    torch.cuda.empty_cache()


def test_rccl_world_size_one_gather_and_image_token_on_device():
    """The only data-path collective (dist.all_gather_shards) and the same-image guard (one all_reduce) executed on DEVICE tensors
    through RCCL (backend 'nccl'), world_size 1 -- the multi-rank logic is covered by the 2-rank gloo tests (tests/test_dist_cpu.py);
    this one proves the RCCL path initialises and runs on this software stack.  Own process: a process group must not leak into
    the other tests."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys, random, torch, torch.distributed as dist
sys.path.insert(0, %r)
from patchfusion_amd.config import make_config
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29653', world_size=1, rank=0, device_id=torch.device('cuda', 0))
cfg = make_config('vits', (112, 154), (448, 616), (2, 2))
sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
img = torch.rand(1, 3, 448, 616, generator=torch.Generator().manual_seed(1234)).cuda()
outs = []
for shard in (False, True):
    m = PatchFusion(cfg, compute_dtype='fp32', shard_patches=shard).eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    random.seed(5621)
    d, _ = m(mode='infer', image_lr=m.resizer(img), image_hr=img, cai_mode='r4', process_num=2)
    outs.append(d.clone())
same = m._same_image_token(m.resizer(img), img)
assert bool(same)
from patchfusion_amd.dist import all_gather_shards
p = torch.randn(5, 7, 9, device='cuda')
assert torch.equal(all_gather_shards(p, 5, 1), p)
assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
dist.destroy_process_group()
print('rccl-ok')
""" % root
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_free_parameters_keeps_the_numbers_and_refuses_what_it_cannot_do():
    """PatchFusion.free_parameters() (opt-in, round 5): once the engine holds every layer in packed form the unpacked checkpoint tensors are released;
    forward results stay BIT-identical, state_dict() / .to() refuse with a clear message, load_state_dict() restores the normal state."""
    cfg, sd, m, img = build(*TINY, "fp32")
    lr = m.resizer(img).cuda()
    with torch.no_grad():
        d0, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode="m1", process_num=2)
    before = sum(p.numel() for p in m.parameters())
    m.free_parameters()
    assert sum(p.numel() for p in m.parameters()) == 0 and before > 0
    with torch.no_grad():
        d1, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode="m1", process_num=2)
    assert torch.equal(d0, d1)
    with pytest.raises(RuntimeError, match="free_parameters"):
        m.state_dict()
    with pytest.raises(RuntimeError, match="free_parameters"):
        m.cpu()
    m.load_state_dict(sd, strict=True)                          # back to the normal state: parameters on the device again, engine rebuilt on demand
    assert sum(p.numel() for p in m.parameters()) == before and next(m.parameters()).is_cuda
    with torch.no_grad():
        d2, _ = m(mode="infer", image_lr=lr, image_hr=img.cuda(), cai_mode="m1", process_num=2)
    assert torch.equal(d0, d2) and len(m.state_dict()) == len(sd)
