"""BaselinePretrain (single-branch models): oracle vs the reference live (build container),
engine wiring (torch reference ops) vs oracle, checkpoint surface."""
import random

import pytest
import torch

from oracle import pf_oracle, ref_shim
from patchfusion_amd.baseline import BaselinePretrain
from patchfusion_amd.config import zoe_branch_config
from patchfusion_amd.spec import branch_spec, synthetic_state_dict
from tests.fake_ops import ops as fake_ops
from collections import OrderedDict

PS, RAW, SPLIT = (112, 154), (448, 616), (2, 2)


def _setup(target):
    bc = zoe_branch_config("vits", PS)
    spec = OrderedDict()
    branch_spec(spec, f"{target}_branch.", bc)
    sd = synthetic_state_dict(spec, 0)
    img = torch.rand(1, 3, *RAW, generator=torch.Generator().manual_seed(1234))
    return bc, sd, img


@pytest.mark.parametrize("target,mode", [("coarse", "m1"), ("fine", "m1"), ("fine", "r1")])
def test_engine_wiring_matches_oracle(target, mode):
    bc, sd, img = _setup(target)
    m = BaselinePretrain(bc, bc, dict(type="SILogLoss"), 1e-3, 80, RAW, PS, SPLIT, target=target, ops=fake_ops).eval()
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_dict({k[len(target) + 8:]: v for k, v in sd.items()})          # checkpoints carry un-prefixed keys
    assert set(m.get_save_dict()) == {k[len(target) + 8:] for k in sd}
    lr = m.resizer(img)
    random.seed(5621)
    d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=mode, process_num=2)
    random.seed(5621)
    o = pf_oracle.BaselineOracle(bc, PS, RAW, SPLIT, sd, target).infer(lr, img, mode, 2)
    assert d.shape == o.shape and (d - o).abs().max() < 2e-5


@pytest.mark.reference
@pytest.mark.parametrize("target,mode", [("coarse", "m1"), ("fine", "m2"), ("fine", "r1")])
def test_oracle_matches_reference_live(target, mode):
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present")
    ref_shim.import_reference()
    from estimator.models.baseline_pretrain import BaselinePretrain as RefBaseline
    bc, sd, img = _setup(target)
    cfg = ref_shim._AttrDict(bc)
    with ref_shim.in_reference_cwd():
        m = RefBaseline(cfg, cfg, dict(type="SILogLoss"), 1e-3, 80, RAW, PS, SPLIT, target=target).eval()
    print(m.load_dict({k[len(target) + 8:]: v for k, v in sd.items()}))
    lr = m.resizer(img)
    with torch.no_grad():
        random.seed(5621)
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, depth_gt=None, cai_mode=mode, process_num=2)
    random.seed(5621)
    o = pf_oracle.BaselineOracle(bc, PS, RAW, SPLIT, sd, target).infer(lr, img, mode, 2)
    assert d.shape == o.shape and (d - o).abs().max() < 1e-5
