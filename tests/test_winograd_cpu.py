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


@pytest.mark.parametrize("sw", [8, 4])
@pytest.mark.parametrize("case", [
    (1, 9, 37, 16, 48, True, False, False),       # several super-tiles, channel half 1 has one valid 16-channel group
    (2, 13, 41, 16, 32, False, True, True),       # two images; upper channel half idle; relu_in + residual
    (1, 6, 150, 16, 96, True, False, False),      # two channel blocks, one long tile row
    (1, 37, 5, 32, 64, False, False, True),       # narrower than one super-tile: mostly padding tiles
])
def test_fused_kernel_index_model_equals_direct_convolution(case, sw):
    """csrc/wino_fused.hip replayed lane by lane in numpy (tests/wino_fused_model.py: same constants and per-lane expressions for the
    super-tile order, the swizzled DMA slots, the transform lanes, the V image, the MFMA fragments against packing.winograd_filters_fused, the
    accumulator layout and the epilogue exchange) must reproduce F.conv2d -- the part of the kernel that can be checked without a GPU."""
    from tests import wino_fused_model as wm
    B, H, W, cin, cout, relu, relu_in, with_res = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, cin, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g, dtype=torch.float64)
    res = torch.randn(B, H, W, cout, generator=g, dtype=torch.float64) if with_res else None
    rows = pk.round_up(pk.round_up(cout, 4), 16)
    wk = torch.zeros(rows, 3, 3, cin)
    wk[:cout] = w.float().permute(0, 2, 3, 1)
    up = pk.winograd_filters_fused(wk)
    assert tuple(up.shape) == ((cout + 63) // 64, cin // 8, 36, 2, 64, 4)
    bp = np.zeros(rows)
    bp[:cout] = bias.numpy()
    conflicts = []
    y = wm.run(x.numpy(), up.numpy(), bp, relu, relu_in, res.numpy() if with_res else None, None, cout, gs=2, SW=sw, conflicts=conflicts)
    assert max(conflicts) == 1                                       # the pixel-pair swizzle keeps the transform's LDS reads conflict free
    xin = x.clamp(min=0) if relu_in else x
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = ref.clamp(min=0)
    if with_res:
        ref = ref + res
    assert not np.isnan(y).any()                                     # every output written exactly by some (strip, channel block)
    assert float(np.abs(y - ref.numpy()).max()) < 2e-5               # float32 rounding of the packed filters only


def test_fused_filters_are_the_three_step_filters_in_fragment_order(wino_env):
    wino_env(4)
    g = torch.Generator().manual_seed(3)
    w = torch.randn(96, 128, 3, 3, generator=g)
    pw = pk.pack_conv(w, None, dtype=torch.float32)
    assert pw.wino_up is not None and tuple(pw.wino_up.shape) == (2, 16, 36, 2, 64, 4)
    U = pw.wino_u                                                     # [36, rows, Kpad]
    for nb, kc, plane, half, lane, e in ((0, 0, 0, 0, 0, 0), (1, 15, 35, 0, 63, 3), (0, 7, 17, 1, 37, 2), (1, 3, 5, 0, 21, 1)):
        r, gq, cg, ks = lane & 15, lane >> 4, e >> 1, e & 1
        n, c = 64 * nb + 32 * half + 16 * cg + r, 8 * kc + 2 * gq + ks
        want = float(U[plane, n, c]) if n < U.shape[1] else 0.0
        assert float(pw.wino_up[nb, kc, plane, half, lane, e]) == want


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
