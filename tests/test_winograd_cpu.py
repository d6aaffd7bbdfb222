"""Host side of the float32 Winograd path (patchfusion_amd/packing.py winograd_filters / winograd_applies) and the torch emulation of
csrc/winograd.hip that the CPU wiring tests run (tests/fake_ops.py _conv_winograd_ref): F(2x2,3x3) and F(4x4,3x3) must reproduce the
direct 3x3 convolution of the same packed layer up to float32 rounding.  The HIP kernels themselves: tests/op_checks.py conv_winograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from patchfusion_amd import packing as pk
from tests.fake_ops import ops as fake


@pytest.fixture
def wino_env(monkeypatch):
    def set_m(m, min_pixels=0):
        monkeypatch.setenv("PF_FAKE_WINOGRAD", "1")
        monkeypatch.setenv("PF_WINOGRAD", str(m))
        monkeypatch.setenv("PF_WINOGRAD_MIN_PIXELS", str(min_pixels))
    return set_m


@pytest.mark.parametrize("m", [2, 4])
def test_filters_are_G_g_Gt(wino_env, m):
    wino_env(m)
    g = torch.Generator().manual_seed(m)
    w = torch.randn(160, 128, 3, 3, generator=g)
    pw = pk.pack_conv(w, torch.zeros(160), dtype=torch.float32)
    a = m + 2
    assert pw.wino_m == m and tuple(pw.wino_u.shape) == (a * a, 160, 128) and pw.wino_u.dtype == torch.float32
    G = np.array(pk.WINO_G[m])
    for o, c in ((0, 0), (159, 127), (77, 5)):
        ref = G @ w[o, c].double().numpy() @ G.T
        got = pw.wino_u[:, o, c].reshape(a, a).double().numpy()
        assert np.abs(got - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max())
    # the transform pair is a valid Winograd algorithm: A^T [(G g G^T) . (B^T d B)] A == correlation of d with g (float64)
    Bt, At = np.array(pk.WINO_BT[m], float), np.array(pk.WINO_AT[m], float)
    d = np.random.RandomState(0).randn(a, a)
    gk = np.random.RandomState(1).randn(3, 3)
    y = At @ ((G @ gk @ G.T) * (Bt @ d @ Bt.T)) @ At.T
    ref = np.array([[(d[i:i + 3, j:j + 3] * gk).sum() for j in range(m)] for i in range(m)])
    assert np.abs(y - ref).max() < 1e-10


def test_eligibility(wino_env):
    wino_env(4)
    w = torch.randn(128, 128, 3, 3)
    assert pk.pack_conv(w, None, dtype=torch.float32).wino_u is not None
    assert pk.pack_conv(w, None, dtype=torch.bfloat16).wino_u is None                       # float32 mode only
    assert pk.pack_conv(w[:, :64], None, dtype=torch.float32).wino_u is None                # K too short
    assert pk.pack_conv(w[:32], None, dtype=torch.float32).wino_u is not None               # 32 output channels still pay
    assert pk.pack_conv(w[:24], None, dtype=torch.float32).wino_u is None                   # not a whole 128-byte chunk
    assert pk.pack_conv(w[:, :, :1, :1], None, dtype=torch.float32).wino_u is None          # 1x1
    assert pk.pack_conv(w, None, dtype=torch.float32, scale=torch.ones(128)).wino_u is None
    pw = pk.pack_conv(w, None, dtype=torch.float32)
    assert pk.winograd_applies(pw, 10, 1, 1, "relu") and not pk.winograd_applies(pw, 10, 2, 1, None)
    assert not pk.winograd_applies(pw, 10, 1, 1, "gelu") and not pk.winograd_applies(pw, 10, 1, 0, None)
    wino_env(4, 100000)
    assert not pk.winograd_applies(pw, 99999, 1, 1, None) and pk.winograd_applies(pw, 100000, 1, 1, None)
    wino_env(0)
    assert pk.pack_conv(w, None, dtype=torch.float32).wino_u is None                        # PF_WINOGRAD=0: direct kernels only
    wino_env(3)
    with pytest.raises(ValueError):
        pk.pack_conv(w, None, dtype=torch.float32)


@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("shape", [(2, 13, 18), (1, 16, 16), (1, 5, 3)])
def test_three_step_path_equals_direct_convolution(wino_env, m, shape):
    wino_env(m)
    B, H, W = shape
    g = torch.Generator().manual_seed(7)
    w = torch.randn(160, 128, 3, 3, generator=g) / (9 * 128) ** 0.5
    b = torch.randn(160, generator=g)
    pw = pk.pack_conv(w, b, dtype=torch.float32)
    xbuf = torch.randn(B, H, W, 136, generator=g)                     # the layer reads a 128-channel slice of a wider buffer
    res = torch.randn(B, H, W, 160, generator=g)
    res2 = torch.randn(B, H, W, 168, generator=g)
    for kw in (dict(), dict(act="relu"), dict(relu_in=True, res=res), dict(act="relu", res=res, res2=res2[..., :160])):
        y1 = torch.full((B, H, W, 176), float("nan"))
        y2 = torch.full((B, H, W, 176), float("nan"))
        fake.conv(xbuf[..., :128], pw, y1[..., 8:168], pad=1, **kw)
        fake.conv(xbuf[..., :128], pw, y2[..., 8:168], pad=1, _direct=True, **kw)
        assert torch.isnan(y1[..., :8]).all() and torch.isnan(y1[..., 168:]).all()
        err = float((y1[..., 8:168] - y2[..., 8:168]).abs().max() / y2[..., 8:168].abs().max())
        assert err <= (1.5e-5 if m == 4 else 3e-6), (m, shape, sorted(kw), err)      # relative to max |y|: float32 rounding only
    # and against torch's own convolution
    ref = F.conv2d(xbuf[..., :128].permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    y = torch.empty(B, H, W, 160)
    fake.conv(xbuf[..., :128], pw, y, pad=1)
    assert float((y - ref).abs().max() / ref.abs().max()) <= (1.5e-5 if m == 4 else 3e-6)


@pytest.mark.parametrize("m", [2, 4])
def test_engine_end_to_end_through_the_winograd_path(wino_env, m, golden_dir):
    """the whole tiny pass with every eligible 3x3 layer (fusion U-Net, 256 channels) on the emulated three-step path: same
    reference golden, same tolerance as the direct path (tests/test_engine_cpu.py)"""
    import os
    from patchfusion_amd.config import make_config
    from patchfusion_amd.model import PatchFusion
    from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
    wino_env(m)
    g = np.load(os.path.join(golden_dir, "tiny_vits.npz"))
    cfg = make_config("vits", (112, 154), (448, 616), (2, 2))
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    mod = PatchFusion(cfg, compute_dtype="fp32", ops=fake).eval()
    mod.load_state_dict(sd, strict=True)
    img = torch.rand(1, 3, 448, 616, generator=torch.Generator().manual_seed(1234))
    d, _ = mod(mode="infer", image_lr=mod.resizer(img), image_hr=img, cai_mode="m1", process_num=2)
    n_wino = sum(1 for pc in _packed(mod._engine) if pc.wino_u is not None and pc.wino_m == m)     # the engine is built on first use
    assert n_wino >= 10, n_wino
    ref = g["depth_m1"]
    assert np.abs(d[0, 0].numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def _packed(mod):
    """every PackedConv the engine holds (walks the engine objects' attributes)"""
    seen, out, stack = set(), [], [mod]
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if isinstance(o, pk.PackedConv):
            out.append(o)
        elif isinstance(o, (list, tuple)):
            stack.extend(o)
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif hasattr(o, "__dict__") and not isinstance(o, torch.Tensor):
            stack.extend(vars(o).values())
    return out
