"""BASELINE configs[4] (ZoeDepth-N PatchFusion: MiDaS/BEiT core + metric-bins head, 384x512 geometry, `r<N>` random tiles),
the encoder-independent part: engine wiring (torch stand-in ops, tests/fake_ops.py) against the oracle with the SAME stand-in
relative-depth core injected on both sides (tests/zoe_core_standin.py; the core itself is PARITY UNPINNED)."""
import random

import pytest
import torch

from oracle import pf_oracle
from patchfusion_amd import tiling
from patchfusion_amd.config import make_zoe_config, pyramid_sizes
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
from tests.fake_ops import ops as fake_ops
from tests.zoe_core_standin import StandInCore

PS, RAW, SPLIT = (96, 128), (384, 512), (2, 2)


@pytest.fixture(scope="module")
def zoe():
    cfg = make_zoe_config(PS, RAW, SPLIT)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    cores = (StandInCore(11), StandInCore(12))
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops, core_providers=cores).eval()
    m.load_state_dict(sd, strict=True)
    img = torch.rand(1, 3, *RAW, generator=torch.Generator().manual_seed(1234))
    return cfg, sd, cores, m, img


def test_zoe_geometry_and_schema():
    cfg = make_zoe_config()                                    # the shipped 384x512 config: fusion geometry from the class defaults
    assert "in_channels" not in cfg["guided_fusion"]
    spec = patchfusion_spec(cfg)
    assert spec["guided_fusion.g2l_list.0.absolute_pos_embed"].shape == (1, 12 * 16, 256)
    assert spec["guided_fusion.g2l_list.5.absolute_pos_embed"].shape == (1, 384 * 512, 32)
    assert spec["coarse_branch.conv2.weight"].shape == (256, 256, 1, 1) and "coarse_branch.core.core.pretrained.cls_token" not in spec
    assert pyramid_sizes((384, 512), "ZoeDepth") == [(384, 512), (192, 256), (96, 128), (48, 64), (24, 32), (12, 16)]
    m = PatchFusion(cfg, ops=fake_ops)
    # midas.py:171-173 / patchfusion.py:84: Resize(..., ensure_multiple_of=32)
    assert m.resizer.m == 32 and m.resizer.get_size(3840, 2160) == (512, 384)
    assert tuple(m.resizer(torch.zeros(1, 3, 540, 960)).shape) == (1, 3, 384, 512)
    # docs/user_infer.md: 4x4 split, r128 -> 16 + 33 + 128 = 177 patches
    random.seed(0)
    assert len(tiling.tile_schedule(m.tile_cfg, m.patch_process_shape, "r128", 4)) == 177
    with pytest.raises(NotImplementedError, match="relative-depth core"):
        m(mode="infer", image_lr=torch.zeros(1, 3, 384, 512), image_hr=torch.zeros(1, 3, 2160, 3840))


@pytest.mark.parametrize("mode", ["m1", "r4"])
def test_zoe_patchfusion_matches_oracle_with_injected_core(zoe, mode):
    cfg, sd, cores, m, img = zoe
    lr = m.resizer(img)
    random.seed(5621)
    with torch.no_grad():
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=mode, process_num=2)
    random.seed(5621)
    ref = pf_oracle.Oracle(cfg, sd, core_providers=cores).infer(lr, img, mode, 2)
    assert d.shape == ref.shape
    assert float((d - ref).abs().max()) < 2e-5, float((d - ref).abs().max())
    assert float(ref.std()) > 1e-3                             # the map is not trivially constant


def test_zoe_branch_features_match_oracle(zoe):
    cfg, sd, cores, m, img = zoe
    lr = m.resizer(img)
    depth, feats = m.coarse_forward(lr)
    od, of = pf_oracle.branch_forward_external(sd, "coarse_branch.", lr, cfg["coarse_branch"], cores[0])
    assert float((depth - od).abs().max()) < 1e-5
    for a, b in zip(feats, of):
        assert a.shape == b.shape and float((a - b).abs().max()) < 1e-5


@pytest.mark.parametrize("target,mode", [("coarse", "m1"), ("fine", "m2"), ("fine", "r1")])
def test_zoe_baseline_pretrain_matches_oracle_with_injected_core(target, mode):
    """BaselinePretrain with a type-'ZoeDepth' branch (baseline_pretrain.py:67-86): multiple-of-32 resizer, external core through
    the provider hook, the rest on the engine -- against BaselineOracle with the same stand-in core."""
    from collections import OrderedDict

    from patchfusion_amd.baseline import BaselinePretrain
    from patchfusion_amd.config import zoe_midas_branch_config
    from patchfusion_amd.spec import branch_spec
    bc = zoe_midas_branch_config(PS)
    spec = OrderedDict()
    branch_spec(spec, f"{target}_branch.", bc)
    sd = synthetic_state_dict(spec, 0)
    core = StandInCore(21)
    m = BaselinePretrain(bc, bc, dict(type="SILogLoss"), 1e-3, 80, RAW, PS, SPLIT, target=target, ops=fake_ops, core_provider=core).eval()
    assert list(m.state_dict().keys()) == list(sd.keys()) and m.resizer.m == 32
    m.load_dict({k[len(target) + 8:]: v for k, v in sd.items()})
    img = torch.rand(1, 3, *RAW, generator=torch.Generator().manual_seed(4321))
    lr = m.resizer(img)
    random.seed(5621)
    d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=mode, process_num=2)
    random.seed(5621)
    o = pf_oracle.BaselineOracle(bc, PS, RAW, SPLIT, sd, target, core_provider=core).infer(lr, img, mode, 2)
    assert d.shape == o.shape and float((d - o).abs().max()) < 2e-5
    with pytest.raises(NotImplementedError, match="relative-depth core"):
        b = BaselinePretrain(bc, bc, None, 1e-3, 80, RAW, PS, SPLIT, target=target, ops=fake_ops)
        b.load_dict({k[len(target) + 8:]: v for k, v in sd.items()})
        b(mode="infer", image_lr=lr, image_hr=img)


def test_zoe_branch_checkpoint_with_core_weights_loads_through_the_configdict_route(tmp_path):
    """A real ZoeDepth branch checkpoint carries the MiDaS/BEiT weights under `core.` (they are the external provider's, not this
    module's): the strict per-branch load of the mmengine ConfigDict route (patchfusion.py:105-109) must accept them -- and still
    reject genuinely unknown / missing keys (round-2 advisor finding: BASELINE configs[4] could not be constructed via tools/test.py)."""
    cfg = make_zoe_config(PS, RAW, SPLIT)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    paths = []
    for br in ("coarse_branch.", "fine_branch."):
        bsd = {k[len(br):]: v for k, v in sd.items() if k.startswith(br)}
        bsd["core.core.pretrained.model.blocks.0.attn.qkv.weight"] = torch.zeros(3, 3)     # the external core's own tensors
        bsd["core.core.scratch.refinenet1.out_conv.bias"] = torch.zeros(4)
        path = str(tmp_path / (br + "pth"))
        torch.save({"model_state_dict": bsd}, path)
        paths.append(path)

    class ConfigDict(dict):          # stands in for mmengine.ConfigDict (model._is_mmengine_configdict checks the class name)
        def to_dict(self):
            return dict(self)

    c = ConfigDict(cfg)
    c["pretrain_model"] = paths
    with pytest.warns(UserWarning, match="kept and handed to the provider"):   # the stand-in cores cannot take weights: said aloud, not dropped silently
        m = PatchFusion(c, ops=fake_ops, core_providers=(StandInCore(11), StandInCore(12)))
    assert m.config.load_branch is True

    class TakingCore(StandInCore):       # a provider that owns its weights receives the `core.` sub-dict, strictly
        def load_state_dict(self, sd, strict=True):
            assert strict and sorted(sd) == ["core.pretrained.model.blocks.0.attn.qkv.weight", "core.scratch.refinenet1.out_conv.bias"]
            self.loaded = dict(sd)

    taking = (TakingCore(11), TakingCore(12))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        PatchFusion(c, ops=fake_ops, core_providers=taking)
    assert all(len(t.loaded) == 2 for t in taking)
    # round-4 advisor finding: providers injected AFTER construction still receive the checkpoint's `core.` weights (they were kept, not dropped)
    late = (TakingCore(11), TakingCore(12))
    m.set_core_providers(*late)
    assert all(len(t.loaded) == 2 for t in late) and m._pending_core_sd == [None, None]
    # ... and a branch checkpoint that fails its own strict check must not have touched the injected core
    untouched = (TakingCore(11), TakingCore(12))
    broken = torch.load(paths[1])
    broken["model_state_dict"].pop("conv2.weight")
    bp = str(tmp_path / "fine_missing.pth")
    torch.save(broken, bp)
    c2 = ConfigDict(cfg)
    c2["pretrain_model"] = [paths[0], bp]
    with pytest.raises(RuntimeError, match="Missing"):
        PatchFusion(c2, ops=fake_ops, core_providers=untouched)
    assert not hasattr(untouched[1], "loaded")
    assert not hasattr(untouched[0], "loaded")       # round-5 advisor finding: BOTH checkpoints are checked before EITHER provider is touched
    # pending `core.` tensors (no provider could take them) can be dropped explicitly instead of living as long as the model
    with pytest.warns(UserWarning, match="kept and handed to the provider"):
        m2 = PatchFusion(c, ops=fake_ops, core_providers=(StandInCore(11), StandInCore(12)))
    assert m2._pending_core_sd[0] is not None and m2._pending_core_sd[1] is not None
    m2.drop_pending_core_weights()
    assert m2._pending_core_sd == [None, None]
    got = m.state_dict()
    assert all(torch.equal(got[k], v) for k, v in sd.items() if k.startswith(("coarse_branch.", "fine_branch.")))
    # an unknown non-core key is still an error, like load_state_dict(strict=True)
    bad = torch.load(paths[0])
    bad["model_state_dict"]["conv2.bogus"] = torch.zeros(1)
    torch.save(bad, paths[0])
    with pytest.raises(RuntimeError, match="Unexpected"):
        PatchFusion(c, ops=fake_ops, core_providers=(StandInCore(11), StandInCore(12)))
