"""`PatchFusion.forward(mode='train')` (patchfusion.py:372-399) -- FORWARD VALUE of the training step
(coarse branch on a batch of images, fine branch on one crop per image, coarse_postprocess_train, fusion_forward, SILogLoss).
  * the oracle restatement (pf_oracle.Oracle.train_forward) against the reference's own Python (build container only),
  * the engine wiring (torch stand-in ops) against the oracle.
The HIP kernels of the same path are checked in tests/test_e2e_gpu.py::test_train_mode_forward_vs_oracle."""
import pytest
import torch

from oracle import pf_oracle
from patchfusion_amd.config import make_config
from patchfusion_amd.model import PatchFusion
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
from tests.fake_ops import ops as fake_ops

TINY = ("vits", (112, 154), (448, 616), (2, 2))


def train_batch(B=2, seed=7):
    g = torch.Generator().manual_seed(seed)
    image_lr = torch.rand(B, 3, 112, 154, generator=g)
    crops = torch.rand(B, 3, 112, 154, generator=g)
    # random crop boxes of the raw patch size (224 x 308) inside the 448 x 616 image, (x1, y1, x2, y2)
    x0 = torch.randint(0, 616 - 308, (B,), generator=g)
    y0 = torch.randint(0, 448 - 224, (B,), generator=g)
    bboxs = torch.stack([x0, y0, x0 + 308, y0 + 224], dim=1)
    gt = 0.3 + torch.rand(B, 1, 112, 154, generator=g)
    gt[:, :, :5] = 0.0                                   # invalid rows (masked by min_depth)
    return image_lr, crops, bboxs, gt


@pytest.fixture(scope="module")
def tiny():
    cfg = make_config(*TINY)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    return cfg, sd


def test_engine_train_forward_matches_oracle(tiny):
    cfg, sd = tiny
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops).eval()
    m.load_state_dict(sd, strict=True)
    image_lr, crops, bboxs, gt = train_batch()
    loss_dict, aux = m(mode="train", image_lr=image_lr, image_hr=None, crops_image_hr=crops, crop_depths=gt, bboxs=bboxs)
    ref_loss, ref_pred = pf_oracle.Oracle(cfg, sd).train_forward(image_lr, crops, gt, bboxs)
    assert set(loss_dict) == {"sig_loss", "total_loss"} and set(aux) == {"rgb", "depth_pred", "depth_gt"}
    assert aux["depth_pred"].shape == ref_pred.shape == (2, 1, 112, 154)
    assert float((aux["depth_pred"] - ref_pred).abs().max()) < 2e-5
    assert abs(float(loss_dict["total_loss"]) - float(ref_loss)) < 1e-4 * max(1.0, abs(float(ref_loss)))
    assert float(ref_loss) > 0.1                           # a real loss value, not the degenerate branch


def test_silog_degenerate_mask_returns_zero(tiny):
    pred = torch.rand(1, 1, 8, 8) + 0.5
    gt = torch.zeros(1, 1, 8, 8)
    assert float(fake_ops.silog_loss(pred, gt, 1e-3, 80)) == 0.0
    assert float(pf_oracle.silog_loss(pred, gt, 1e-3, 80).abs().max()) == 0.0


@pytest.mark.reference
def test_oracle_train_forward_matches_reference_live(tiny):
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("reference tree not present")
    cfg, sd = tiny
    PF = ref_shim.import_reference()
    with ref_shim.in_reference_cwd():
        ref = PF(cfg).eval()
    ref.load_state_dict(sd, strict=True)
    image_lr, crops, bboxs, gt = train_batch()
    with torch.no_grad():
        loss_dict, aux = ref(mode="train", image_lr=image_lr, image_hr=None, crops_image_hr=crops, crop_depths=gt, bboxs=bboxs)
    o_loss, o_pred = pf_oracle.Oracle(cfg, sd).train_forward(image_lr, crops, gt, bboxs)
    assert float((aux["depth_pred"] - o_pred).abs().max()) < 1e-5
    assert abs(float(loss_dict["total_loss"]) - float(o_loss)) < 1e-5 * max(1.0, abs(float(o_loss)))
