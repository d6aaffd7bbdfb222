"""TEST INFRASTRUCTURE: per-op parity checks of the hand-written HIP kernels (through the C ABI,
patchfusion_amd.hip_ops) against the plain-PyTorch fp32 reference of the same op (tests/fake_ops.py).

Each check is ``name -> callable(dtype) -> (err, tol, info)`` where err = max|hip - ref| / max(1, max|ref|).
Used by tests/test_hip_ops_gpu.py (pytest -m gpu) and tools/gpu_selfcheck.py (one subprocess per
check so that a faulting kernel cannot hide the others).
"""
import torch

from patchfusion_amd import packing as pk
from tests.fake_ops import ops as ref_ops

DEV = "cuda"


def hip():
    from patchfusion_amd.hip_ops import ops
    return ops


def _switches_changed():
    """hip_ops resolves its PF_* switches and per-layer dispatch plans once (refresh_env): checks that flip a switch between calls say so"""
    from patchfusion_amd import hip_ops
    hip_ops.refresh_env()


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def _tol(dtype, f32=2e-4, bf16=2e-2):
    return f32 if dtype == torch.float32 else bf16


def _err(a, b):
    a, b = a.float(), b.float()
    if not torch.isfinite(a).all():
        return float("inf")
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


_FLUSH = []


def _flush_caches():
    """overwrite 1 GiB: L2 (4 MiB per XCD) and the 256 MiB memory-side cache forget the operands of the next launch"""
    if not _FLUSH:
        _FLUSH.append(torch.empty(256 * 2 ** 20, dtype=torch.float32, device=DEV))
    _FLUSH[0].add_(1.0)


def _conv_case(dtype, B, H, W, cin, cout, k, stride=1, pad=0, act=None, relu_in=False, res=False, res2=False, bias=True,
               scale=False, x_extra=0, y_extra=0, out_f32=False, inplace=False, cin_real=None, seed=0):
    cin_real = cin_real or cin
    w = torch.randn(cout, cin_real, k, k, generator=torch.Generator().manual_seed(seed)) / (cin_real * k * k) ** 0.5
    b = torch.randn(cout, generator=torch.Generator().manual_seed(seed + 1)) if bias else None
    sc = (0.5 + torch.rand(cout, generator=torch.Generator().manual_seed(seed + 2))) if scale else None
    pw = pk.pack_conv(w, b, dtype=dtype, cin_total=cin, scale=sc).to(DEV)
    xb = _rand((B, H, W, cin + x_extra), dtype, seed + 3)
    x = xb[..., x_extra // 2: x_extra // 2 + cin] if x_extra else xb
    if x_extra and (x_extra // 2) % 8:
        raise ValueError("x_extra/2 must be a multiple of 8")
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    ydt = torch.float32 if out_f32 else dtype
    outs = []
    r1 = _rand((B, OH, OW, pw.cout), dtype, seed + 4) if res else None
    r2 = _rand((B, OH, OW, pw.cout), dtype, seed + 5) if res2 else None
    for o in (hip(), ref_ops):
        yb = torch.zeros((B, OH, OW, pw.cout + y_extra), dtype=ydt, device=DEV)
        y = yb[..., y_extra // 2: y_extra // 2 + pw.cout] if y_extra else yb
        rr = r1
        if inplace:
            y.copy_(r1)
            rr = y
        o.conv(x, pw, y, stride=stride, pad=pad, act=act, relu_in=relu_in, res=rr, res2=r2)
        outs.append(yb.clone())
    torch.cuda.synchronize()
    return _err(outs[0], outs[1]), _tol(dtype), f"conv B{B} {H}x{W} {cin}->{cout} k{k}"


def conv_winograd(dt):
    """float32 Winograd F(2x2,3x3) / F(4x4,3x3) (csrc/winograd.hip: input transform -> one batched GEMM launch over the (m+2)^2 planes
    -> output transform + epilogue) against the DIRECT float32 convolution reference, incl. odd sizes (partial tiles), channel-slice
    views, relu_in and residuals.  Tolerance: float32 rounding of the transforms (m = 4: ~15x a direct conv's), relative to max |y|."""
    import os
    old = {k: os.environ.get(k) for k in ("PF_WINOGRAD", "PF_WINOGRAD_MIN_PIXELS")}
    errs = []
    try:
        os.environ["PF_WINOGRAD_MIN_PIXELS"] = "0"
        for m in (2, 4):
            os.environ["PF_WINOGRAD"] = str(m)
            _switches_changed()
            for i, (B, H, W, cin, cout, kw) in enumerate((
                    (2, 37, 41, 128, 160, dict(act="relu")),
                    (1, 64, 64, 256, 128, dict(relu_in=True, res=True, res2=True)),
                    (1, 30, 43, 544, 544, dict()),
                    (3, 5, 3, 128, 128, dict(act="relu", res=True)))):
                g = torch.Generator().manual_seed(100 + i)
                w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
                pw = pk.pack_conv(w, torch.randn(cout, generator=g), dtype=torch.float32).to(DEV)
                assert pw.wino_m == m and pw.wino_u is not None
                xb = _rand((B, H, W, cin + 16), torch.float32, 200 + i)
                x = xb[..., 8:8 + cin]
                r1 = _rand((B, H, W, cout), torch.float32, 300 + i) if kw.get("res") else None
                r2 = _rand((B, H, W, cout + 8), torch.float32, 400 + i)[..., :cout] if kw.get("res2") else None
                outs = []
                for o, direct in ((hip(), None), (ref_ops, True)):
                    yb = torch.zeros((B, H, W, cout + 16), dtype=torch.float32, device=DEV)
                    o.conv(x, pw, yb[..., 8:8 + cout], pad=1, act=kw.get("act"), relu_in=kw.get("relu_in", False), res=r1, res2=r2, _direct=direct)
                    outs.append(yb)
                errs.append(_err(outs[0], outs[1]) / (4.0 if m == 4 else 1.0))     # m = 4 budget: 4x the m = 2 one
                assert float(outs[0][..., :8].abs().max()) == 0.0 and float(outs[0][..., 8 + cout:].abs().max()) == 0.0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _switches_changed()
    torch.cuda.synchronize()
    return max(errs), 8e-6, "winograd F(2,3) / F(4,3) vs direct f32"


def conv_winograd_subbatch(dt):
    """three-step split-precision Winograd layers under the workspace cap (hip_ops.wino3_window: PF_WS_CAP_GB makes a layer run in windows of its
    Winograd tiles through one smaller arena pair, csrc/winograd.hip run_split3; windows that cut through images and rows, a ragged last window) and under the measured scheduling switches of csrc/winograd.hip run_split3 (resident transforms
    PF_W3_TGRID, capped GEMM grid PF_W3_GRID, GEMM token PF_W3_TOKEN): every variant BIT-IDENTICAL to the plain launch (no op mixes images, the
    switches move work between CUs only), incl. channel-slice views and both residuals."""
    import os
    keys = ("PF_WINOGRAD", "PF_WINOGRAD_MIN_PIXELS", "PF_WINO_FUSED", "PF_WS_CAP_GB", "PF_W3_TGRID", "PF_W3_GRID", "PF_W3_TOKEN")
    old = {k: os.environ.get(k) for k in keys}
    worst = 0.0
    try:
        os.environ.update(PF_WINOGRAD="4", PF_WINOGRAD_MIN_PIXELS="0", PF_WINO_FUSED="0")
        for i, (B, H, W, cin, cout, kw) in enumerate(((4, 37, 41, 256, 256, dict(act="relu", res=True, res2=True)),
                                                      (6, 30, 43, 544, 320, dict(relu_in=True)),
                                                      (8, 56, 74, 256, 256, dict(act="relu")))):
            g = torch.Generator().manual_seed(700 + i)
            w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
            pw = pk.pack_conv(w, torch.randn(cout, generator=g), dtype=torch.float32).to(DEV)
            assert pw.wino_u3 is not None
            x = _rand((B, H, W, cin + 16), torch.float32, 710 + i)[..., 8:8 + cin]
            r1 = _rand((B, H, W, cout), torch.float32, 720 + i) if kw.get("res") else None
            r2 = _rand((B, H, W, cout + 8), torch.float32, 730 + i)[..., :cout] if kw.get("res2") else None
            outs = []
            for env in (dict(PF_WS_CAP_GB="100"), dict(PF_WS_CAP_GB="0.02"), dict(PF_WS_CAP_GB="0.0001"), dict(PF_W3_TGRID="8", PF_W3_GRID="64", PF_W3_TOKEN="1")):
                for k in keys[3:]:
                    os.environ.pop(k, None)
                os.environ.update(env)
                _switches_changed()
                if "PF_WS_CAP_GB" in env:
                    from patchfusion_amd import hip_ops
                    nwin = hip_ops.wino3_window(B, H, W, pw)[1]
                    assert (nwin == 1) == (env["PF_WS_CAP_GB"] == "100") and (env["PF_WS_CAP_GB"] != "0.0001" or nwin >= 2), (env, nwin)
                yb = torch.zeros((B, H, W, cout + 16), dtype=torch.float32, device=DEV)
                hip().conv(x, pw, yb[..., 8:8 + cout], pad=1, act=kw.get("act"), relu_in=kw.get("relu_in", False), res=r1, res2=r2)
                outs.append(yb)
            for o in outs[1:]:
                worst = max(worst, float((o - outs[0]).abs().max()))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _switches_changed()
    torch.cuda.synchronize()
    return worst, 0.0, "three-step Winograd: sub-batched / resident / capped / token-chained == plain launch, bit for bit"


def conv_winograd_fused(dt):
    """fused float32 Winograd F(4x4,3x3) (csrc/wino_fused.hip: LDS-DMA halo -> in-kernel B^T d B -> 36-plane MFMA -> in-kernel A^T M A +
    epilogue) against the DIRECT float32 convolution, and against the three-step path it replaces.  Cases: partial tiles in both
    directions, strips that wrap tile rows and images, W at the 29-pixel limit, Cout % 64 == 32 (idle upper channel half), several channel
    blocks, channel-slice views of x / y / res2, relu_in, residuals, and two group sizes of the block order."""
    import os
    keys = ("PF_WINOGRAD", "PF_WINOGRAD_MIN_PIXELS", "PF_WINO_FUSED", "PF_WINO_GS")
    old = {k: os.environ.get(k) for k in keys}
    errs, errs3 = [], []
    try:
        os.environ["PF_WINOGRAD_MIN_PIXELS"] = "0"
        os.environ["PF_WINOGRAD"] = "4"
        for i, (B, H, W, cin, cout, gs, kw) in enumerate((
                (2, 37, 41, 128, 160, 8, dict(act="relu")),
                (1, 64, 64, 256, 128, 3, dict(relu_in=True, res=True, res2=True)),
                (1, 30, 43, 544, 544, 8, dict()),
                (3, 5, 29, 128, 96, 1, dict(act="relu", res=True)),
                (8, 56, 74, 768, 256, 8, dict(act="relu")),
                (1, 392, 518, 128, 32, 8, dict(act="relu")),
                (2, 112, 148, 256, 256, 5, dict(relu_in=True, act="relu", res=True)),
                # fused-ONLY layers (32 <= Cin < 128 or Cout % 32 != 0: no three-step form, PF_WINO_FUSED=0 runs the direct kernel): odd
                # chunk-pair DMA tails (Cin / 8 odd pairs), partial channel quads in the epilogue, partial tiles
                (1, 37, 45, 48, 64, 8, dict(act="relu")),
                (8, 98, 129, 64, 32, 8, dict(act="relu")),
                (1, 21, 33, 80, 36, 4, dict(res=True)),
                (2, 17, 23, 32, 4, 8, dict(act="relu", relu_in=True)))):
            os.environ["PF_WINO_GS"] = str(gs)
            _switches_changed()
            g = torch.Generator().manual_seed(100 + i)
            w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
            pw = pk.pack_conv(w, torch.randn(cout, generator=g), dtype=torch.float32).to(DEV)
            assert pw.wino_m == 4 and pw.wino_up is not None and (pw.wino_u is None) == (cin < 128 or cout % 32 != 0)
            xb = _rand((B, H, W, cin + 16), torch.float32, 200 + i)
            x = xb[..., 8:8 + cin]
            r1 = _rand((B, H, W, cout), torch.float32, 300 + i) if kw.get("res") else None
            r2 = _rand((B, H, W, cout + 8), torch.float32, 400 + i)[..., :cout] if kw.get("res2") else None
            outs = []
            for o, direct, fused in ((hip(), None, "2"), (hip(), None, "0"), (ref_ops, True, "0")):
                os.environ["PF_WINO_FUSED"] = fused
                _switches_changed()
                yb = torch.zeros((B, H, W, cout + 16), dtype=torch.float32, device=DEV)
                o.conv(x, pw, yb[..., 8:8 + cout], pad=1, act=kw.get("act"), relu_in=kw.get("relu_in", False), res=r1, res2=r2, _direct=direct)
                outs.append(yb)
            errs.append(_err(outs[0], outs[2]))
            errs3.append(_err(outs[0], outs[1]))
            assert float(outs[0][..., :8].abs().max()) == 0.0 and float(outs[0][..., 8 + cout:].abs().max()) == 0.0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _switches_changed()
    torch.cuda.synchronize()
    return max(max(errs), max(errs3)), 3.2e-5, f"fused winograd F(4,3) vs direct f32 (max {max(errs):.2e}) and vs three-step (max {max(errs3):.2e})"


def conv_gemm_qkv(dt):
    return _conv_case(dt, 1, 1, 2 * 1037, 384, 1152, 1)


def conv_gemm_fc2_inplace(dt):
    return _conv_case(dt, 1, 1, 1037, 1536, 384, 1, scale=True, res=True, inplace=True)


def conv_gemm_gelu(dt):
    return _conv_case(dt, 1, 1, 777, 384, 1536, 1, act="gelu")


def conv3x3_rcu(dt):
    return _conv_case(dt, 2, 28, 37, 64, 64, 3, pad=1, relu_in=True, res=True, res2=True)


def conv3x3_stride2(dt):
    return _conv_case(dt, 2, 28, 37, 384, 384, 3, stride=2, pad=1)


def conv3x3_small_cin(dt):
    return _conv_case(dt, 2, 40, 52, 8, 32, 3, pad=1, act="relu", cin_real=5)


def conv3x3_n160_views(dt):
    return _conv_case(dt, 1, 30, 41, 160, 160, 3, pad=1, act="relu", x_extra=32, y_extra=32)


def conv3x3_n544(dt):
    return _conv_case(dt, 1, 24, 31, 544, 544, 3, pad=1, act="relu")


def conv1x1_cout1_f32out(dt):
    return _conv_case(dt, 2, 17, 23, 128, 1, 1, act="softplus", out_f32=True)


def conv1x1_cout80(dt):
    return _conv_case(dt, 1, 33, 47, 168, 80, 1, act="gelu")


def conv1x1_cout16(dt):
    return _conv_case(dt, 1, 33, 47, 128, 16, 1, act="softplus", out_f32=True)


def conv3x3_nobias_48(dt):
    return _conv_case(dt, 1, 32, 44, 48, 64, 3, pad=1, bias=False)


# ---- shapes that take the 8-wave 256x128 "big" bf16 kernel (Cout >= 96 and >= 2048 output pixels) ----
def conv_big_3x3_rcu(dt):
    return _conv_case(dt, 2, 40, 52, 128, 128, 3, pad=1, relu_in=True, res=True, res2=True, seed=11)


def conv_big_3x3_n544_views(dt):
    return _conv_case(dt, 1, 48, 50, 544, 544, 3, pad=1, act="relu", x_extra=32, y_extra=32, seed=12)


def conv_big_3x3_n160(dt):
    return _conv_case(dt, 3, 30, 41, 160, 160, 3, pad=1, act="relu", seed=13)


# ---- 3x3 tap-reuse (halo) kernel variants; PF_HALO_FORCE=1 routes these small shapes to it (bf16 only; the fp32
# ---- run of the same case exercises the generic kernel).  Shapes hang over the tile edges on purpose: rows
# ---- (H % 16 != 0 -> idle wave rows), columns (W % 32 != 0) and channels (Cout % BN != 0 -> skipped fragments).
def _halo_case(*a, **k):
    import os
    old = os.environ.get("PF_HALO_FORCE")
    os.environ["PF_HALO_FORCE"] = "1"
    try:
        return _conv_case(*a, **k)
    finally:
        if old is None:
            del os.environ["PF_HALO_FORCE"]
        else:
            os.environ["PF_HALO_FORCE"] = old


def conv_halo_n32(dt):
    return _halo_case(dt, 2, 40, 52, 64, 32, 3, pad=1, act="relu", y_extra=16, seed=21)


def conv_halo_n64_res(dt):
    return _halo_case(dt, 2, 33, 67, 160, 64, 3, pad=1, relu_in=True, res=True, seed=22)


def conv_halo_n32_cin544(dt):
    return _halo_case(dt, 1, 50, 70, 544, 32, 3, pad=1, act="relu", seed=23)


def conv_halo_bn192_n544(dt):      # <2,4,3>: 544 = 192 + 192 + 160 (one skipped fragment), 40 rows = 2.5 tiles
    return _halo_case(dt, 1, 40, 70, 64, 544, 3, pad=1, act="relu", seed=24)


def conv_halo_bn128_n224(dt):      # <2,4,2>: 224 = 128 + 96 (one skipped fragment), 21 rows (partial + idle wave rows)
    return _halo_case(dt, 2, 21, 45, 96, 224, 3, pad=1, relu_in=True, res=True, res2=True, seed=25)


def conv_halo_bn192_n160(dt):      # <2,4,3> with a single 192 tile: second channel half keeps 2 of 3 fragments
    return _halo_case(dt, 1, 24, 31, 32, 160, 3, pad=1, seed=26)


def conv_big_stride2(dt):
    return _conv_case(dt, 1, 96, 100, 96, 192, 3, stride=2, pad=1, seed=14)


def conv_big_gemm_scale_inplace(dt):
    return _conv_case(dt, 1, 1, 2 * 1037, 1536, 384, 1, scale=True, res=True, inplace=True, seed=15)


def conv_big_gemm_gelu_k1024(dt):
    return _conv_case(dt, 1, 1, 3 * 1037, 1024, 4096, 1, act="gelu", seed=16)


# ---- f32 exact channel split (544 = 5 x 96 + 64 etc.: body + remainder launches, dispatch_generic) and the 64x64 tile ----
def conv_split_n272_res_views(dt):
    return _conv_case(dt, 2, 20, 27, 64, 272, 3, pad=1, relu_in=True, res=True, res2=True, scale=True, y_extra=32, seed=31)


def conv_split_gemm_n544_inplace(dt):
    return _conv_case(dt, 1, 1, 1037, 256, 544, 1, scale=True, res=True, inplace=True, act="gelu", seed=32)


def conv_gemm_vitl_linear_shape(dt):      # M = 8 x 1037 rows like the ViT-L linears (tile choice by the makespan model)
    return _conv_case(dt, 1, 1, 8 * 1037, 256, 1024, 1, scale=True, res=True, seed=33)


def conv_bf16_pp(dt):
    """bf16 linear layers on the 256 x 128 ping-pong tiles (pf_gemm_bf16_pp, opt-in: PF_BF16_PP=1) against the torch reference AND the
    implicit-GEMM kernel (PF_BF16_PP=0): ragged token counts, every epilogue option, in-place residual, float32 output"""
    import os
    worst, info = 0.0, []
    cases = [dict(B=1, H=1, W=8 * 1037, cin=1024, cout=3072, k=1, seed=51), dict(B=1, H=1, W=2500, cin=4096, cout=1024, k=1, scale=True, res=True, inplace=True, seed=52),
             dict(B=1, H=1, W=2049, cin=1024, cout=1024, k=1, res=True, res2=True, act="relu", seed=53), dict(B=1, H=1, W=3 * 1037, cin=1024, cout=4096, k=1, act="gelu", seed=54),
             dict(B=2, H=37, W=41, cin=512, cout=516, k=1, out_f32=True, y_extra=24, x_extra=16, seed=55)]
    for c in cases:
        errs = []
        for flag in ("1", "0"):
            os.environ["PF_BF16_PP"] = flag
            _switches_changed()
            e, tol, _ = _conv_case(dt, c["B"], c["H"], c["W"], c["cin"], c["cout"], c["k"], **{k: v for k, v in c.items() if k not in ("B", "H", "W", "cin", "cout", "k")})
            errs.append(e)
        os.environ.pop("PF_BF16_PP", None)
        _switches_changed()
        worst = max(worst, errs[0])
        info.append(f"{c['cin']}->{c['cout']} M={c['B'] * c['H'] * c['W']}: pp {errs[0]:.2e} igemm {errs[1]:.2e}")
    return worst, _tol(dt), "; ".join(info)


def conv_dominant_launch(dt):
    """THE dominant launch at its real size: 3x3 544->544 @ 8x392x518 (GuidedFusion Upv1, guided_fusion_model.py:85-100)
    against F.conv2d in fp32 on the same (dtype-rounded) operands."""
    return _conv_case(dt, 8, 392, 518, 544, 544, 3, pad=1, act="relu", seed=34)


def conv_big_transpose(dt):
    w = torch.randn(96, 96, 2, 2, generator=torch.Generator().manual_seed(3)) / 96 ** 0.5
    b = torch.randn(96, generator=torch.Generator().manual_seed(4))
    pw = pk.pack_conv_transpose(w, b, dtype=dt).to(DEV)
    x = _rand((2, 40, 52, 96), dt, 7)
    ys = []
    for o in (hip(), ref_ops):
        y = torch.zeros((2, 80, 104, 96), dtype=dt, device=DEV)
        o.conv(x, pw, y)
        ys.append(y)
    torch.cuda.synchronize()
    return _err(ys[0], ys[1]), _tol(dt), "convT s2 big"


def conv_transpose(dt):
    errs = []
    for s, cin in ((4, 48), (2, 96)):
        w = torch.randn(cin, cin, s, s, generator=torch.Generator().manual_seed(s)) / cin ** 0.5
        b = torch.randn(cin, generator=torch.Generator().manual_seed(s + 1))
        pw = pk.pack_conv_transpose(w, b, dtype=dt).to(DEV)
        x = _rand((2, 8, 11, cin), dt, 7)
        ys = []
        for o in (hip(), ref_ops):
            y = torch.zeros((2, 8 * s, 11 * s, cin), dtype=dt, device=DEV)
            o.conv(x, pw, y)
            ys.append(y)
        ref = torch.nn.functional.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.to(dt).float().to(DEV), b.to(DEV), stride=s).permute(0, 2, 3, 1)
        errs += [_err(ys[0], ys[1]), _err(ys[1], ref)]
    torch.cuda.synchronize()
    return max(errs), _tol(dt), "convT s4,s2"


def patch_embed_tokens(dt):
    img = torch.rand(2, 3, 112, 154, generator=torch.Generator().manual_seed(1)).to(DEV)
    outs = []
    cls, pos = _rand((384,), torch.float32, 2), _rand((89, 384), torch.float32, 3)
    emb = _rand((2 * 88, 384), dt, 4)
    for o in (hip(), ref_ops):
        col = torch.zeros((2 * 88, 592), dtype=dt, device=DEV)
        o.patch_im2col(img, col)
        tok = torch.zeros((2, 89, 384), dtype=dt, device=DEV)
        o.assemble_tokens(emb, tok, cls, pos)
        outs.append((col, tok))
    torch.cuda.synchronize()
    return max(_err(outs[0][0], outs[1][0]), _err(outs[0][1], outs[1][1])), _tol(dt, 1e-5, 1e-2), "im2col+tokens"


def layernorm(dt):
    errs = []
    for D in (384, 1024, 32, 64):
        x = _rand((2 * 89, D), dt, D, 2.0)
        g, b = 1 + 0.1 * _rand((D,), torch.float32, 1), 0.1 * _rand((D,), torch.float32, 2)
        ys = []
        for o in (hip(), ref_ops):
            y = torch.zeros((2 * 89, D), dtype=dt, device=DEV)
            o.layernorm(x, y, g, b, 1e-6)
            y2 = torch.zeros((2 * 88, D), dtype=dt, device=DEV)
            o.layernorm(x, y2, g, b, 1e-6, batches=2, in_rows_per_batch=89, in_row_offset=1, out_rows_per_batch=88)
            ys.append(torch.cat([y, y2]))
        errs.append(_err(ys[0], ys[1]))
    torch.cuda.synchronize()
    return max(errs), _tol(dt, 1e-5, 1e-2), "layernorm"


def vit_attention(dt):
    errs = []
    for B, S, H in ((2, 1037, 6), (1, 89, 16), (3, 64, 2)):
        qkv = _rand((B * S, 3 * H * 64), dt, S)
        ys = []
        for o in (hip(), ref_ops):
            y = torch.zeros((B * S, H * 64), dtype=dt, device=DEV)
            o.vit_attention(qkv, y, B, S, H)
            ys.append(y)
        errs.append(_err(ys[0], ys[1]))
    torch.cuda.synchronize()
    return max(errs), _tol(dt, 1e-4, 2e-2), "vit attention"


def vit_attention_split(dt):
    """float32: the round-2 pair pf_qkv_split + pf_vit_attention (PF_ATTN_QKV=0) stays covered next to the version-2 kernel that reads the
    QKV rows directly (the default, checked by vit_attention above)"""
    import os
    old = os.environ.get("PF_ATTN_QKV")
    os.environ["PF_ATTN_QKV"] = "0"
    _switches_changed()
    try:
        return vit_attention(dt)
    finally:
        if old is None:
            os.environ.pop("PF_ATTN_QKV", None)
        else:
            os.environ["PF_ATTN_QKV"] = old
        _switches_changed()


def vit_attention_split3(dt):
    """attention entirely in split precision as the engine calls it (ops.vit_attention on three-plane q / k / v: the pipelined kernel of csrc/attn_split3.hip)
    against float64 on the SAME float32 qkv, next to the f32-MFMA attention kernel's own error on the same case: ragged S (last key block of 13 keys, query
    blocks without queries), a short sequence, large logits.  Criterion per case: error <= 1.25 x max(3e-6, the f32-MFMA kernel's error) -- float32 grade."""
    o = hip()
    g = torch.Generator().manual_seed(77)
    worst, info = 0.0, []
    for (B, S, heads, scale) in ((2, 1037, 16, 1.0), (1, 70, 2, 3.0), (3, 64, 4, 1.0), (1, 129, 1, 6.0)):
        D = heads * 64
        qkv = (torch.randn(B * S, 3 * D, generator=g) * scale).to(DEV)
        q, k, v = qkv.double().view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
        ref = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * S, D)
        den = max(1.0, float(ref.abs().max()))
        q3 = torch.empty(3, B * S, 3 * D, dtype=torch.bfloat16, device=DEV)
        o.split3(qkv, q3)
        o3 = torch.full((3, B * S, D), 7.0, dtype=torch.bfloat16, device=DEV)
        o.vit_attention(q3, o3, B, S, heads)
        e3 = float((o3.double().sum(0) - ref).abs().max()) / den
        of = torch.empty(B * S, D, device=DEV)
        o.vit_attention(qkv, of, B, S, heads)
        ef = float((of.double() - ref).abs().max()) / den
        worst = max(worst, e3 / (1.25 * max(3e-6, ef)))
        info.append(f"B{B} S{S} h{heads} x{scale}: split {e3:.2e} vs f32 kernel {ef:.2e}")
    return worst, 1.0, "; ".join(info)


def vit_attention_split3_v2(dt):
    """version 2 of the split-precision attention (csrc/attn_split3.hip) through the C ABI, every schedule: the two-phase kernel (LDS-DMA tiles, transposing V
    reads, 16 / 32 queries per wave) must equal version 1 (csrc/vit.hip) BIT FOR BIT -- same products, same order --, the pipelined kernel (32 x 32 x 16 MFMAs,
    32-key blocks, deferred rescale) must stay within 1.25 x max(3e-6, the f32-MFMA kernel's error) of float64 on the same float32 qkv (float32 grade: its
    16-deep MFMAs round the accumulator twice as often as version 1's 32-deep ones -- measured 3.2e-6 against the f32 kernel's 3.0e-6 on the large-logit case).  Ragged S (last key block
    of 13 keys, query blocks and waves without queries), S below one block / one tile, S a multiple of 32 and of 64, head counts that are not a multiple of 8
    (block order fallback), large logits, one key far above the rest late in the sequence (the rescale path), row- and chunk-major outputs."""
    from patchfusion_amd.hip_ops import _L, _p, _stream, check
    o = hip()
    g = torch.Generator().manual_seed(78)
    worst, info, same = 0.0, [], True
    for (B, S, heads, scale) in ((2, 1037, 16, 1.0), (1, 70, 2, 3.0), (3, 64, 4, 1.0), (1, 129, 1, 6.0), (1, 13, 2, 1.0), (9, 300, 6, 1.0), (1, 33, 1, 2.0),
                                 (2, 96, 3, 1.0), (1, 32, 2, 1.0), (1, 1037, 8, 4.0)):
        D = heads * 64
        qkv = (torch.randn(B * S, 3 * D, generator=g) * scale).to(DEV)
        if S in (129, 1037) and B == 1:                          # one key far above the rest late in the sequence: the exponent reference jumps (rescale path)
            qkv.view(B, S, 3, heads, 64)[:, S - 29, 1] *= 8.0
        q, k, v = qkv.double().view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
        ref = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * S, D)
        den = max(1.0, float(ref.abs().max()))
        q3 = torch.empty(3, B * S, 3 * D, dtype=torch.bfloat16, device=DEV)
        o.split3(qkv, q3)
        of = torch.empty(B * S, D, device=DEV)
        check(_L.pf_vit_attention_qkv(_p(qkv), _p(of), B, S, heads, 0, _stream()), "f32")
        ef = float((of.double() - ref).abs().max()) / den
        for kmaj in (0, 1):
            shape = (3, D // 32, B * S, 32) if kmaj else (3, B * S, D)
            o1 = torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)
            check(_L.pf_vit_attention_split3(_p(q3), q3.stride(0), _p(o1), o1.stride(0), kmaj, B, S, heads, _stream()), "v1")
            for qw, sched in ((16, 1), (32, 1), (0, 1), (32, 2), (0, 0)):
                o2 = torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)
                check(_L.pf_vit_attention_split3_v2(_p(q3), q3.stride(0), _p(o2), o2.stride(0), kmaj, B, S, heads, qw, sched, _stream()), "v2")
                rowmaj = o2.permute(0, 2, 1, 3).reshape(3, B * S, D) if kmaj else o2
                e = float((rowmaj.double().sum(0) - ref).abs().max()) / den
                eq = bool(torch.equal(o1, o2))
                if sched == 1:                                   # the two-phase schedule evaluates exactly version 1's operations
                    same = same and eq
                worst = max(worst, e / (1.25 * max(3e-6, ef)))
                if kmaj == 0 or not (eq if sched == 1 else e <= 1.25 * max(3e-6, ef)):
                    info.append(f"B{B} S{S} h{heads} x{scale} kmaj{kmaj} qw{qw} sched{sched}: {e:.2e} (f32 kernel {ef:.2e}) identical={eq}")
    return (worst if same else float("inf")), 1.0, "; ".join(info)


def gemm_split3(dt):
    """split-precision linear layer (csrc/gemm_split3.hip) against float64 on the SAME float32 operands: the three-plane split must be
    exact (h + m + l == x bit for bit), and the GEMM's error must be float32-grade (the dropped terms are O(2^-24) of |x||w|), at a
    ragged token count with every epilogue option, f32 output and split (three-plane) output, both tile shapes."""
    import os
    o = hip()
    g = torch.Generator().manual_seed(71)
    errs = []
    x = (torch.randn(1037, 96, generator=g) * torch.logspace(-6, 6, 96)).to(DEV)
    xb = torch.zeros(1037, 128, device=DEV)
    xb[:, 8:104] = x
    x3 = torch.zeros(3, 1037, 96, dtype=torch.bfloat16, device=DEV)
    o.split3(xb[:, 8:104], x3)
    exact = bool((x3.double().sum(0) == x.double()).all())
    h, m, l = pk.split3(x.cpu())
    same = bool((x3.cpu() == torch.stack([h, m, l])).all())
    info = [f"split exact={exact} host-identical={same}"]
    if not (exact and same):
        return float("inf"), 1e-6, info[0]
    for tile, (M, K, N, act, res, res2, scale) in enumerate([(1037, 256, 544, "gelu", True, True, True), (2 * 1037, 1024, 1024, None, True, False, False),
                                                             (777, 1024, 3072, None, False, False, False), (1037, 4096, 1024, "relu", True, False, True)]):
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        sc = (0.5 + torch.rand(N, generator=g)) if scale else None
        pw3 = pk.pack_conv_split3(w, b, scale=sc).to(DEV)
        pw = pk.pack_conv(w.view(N, K, 1, 1), b, dtype=torch.float32, scale=sc).to(DEV)
        x = torch.randn(M, K, generator=g).to(DEV)
        r1 = torch.randn(M, N, generator=g).to(DEV) if res else None
        r2 = torch.randn(M, N, generator=g).to(DEV) if res2 else None
        ref = x.double() @ w.double().t().to(DEV) + b.double().to(DEV)
        if act == "gelu":
            ref = torch.nn.functional.gelu(ref)
        elif act == "relu":
            ref = torch.relu(ref)
        if sc is not None:
            ref = ref * sc.double().to(DEV)
        if r1 is not None:
            ref = ref + r1.double()
        if r2 is not None:
            ref = ref + r2.double()
        den = max(1.0, float(ref.abs().max()))
        x3 = torch.empty(3, M, K, dtype=torch.bfloat16, device=DEV)
        o.split3(x, x3)
        for force in ("64", "128", "persist16", "persist64", "t192_8", "t192_16", "t192_64"):
            # persist<G>: the persistent tile walk (csrc/gemm_split3.hip gemm_split3_persist_kernel) forced on these small shapes with a
            # G-block grid, so that every block crosses several tile boundaries (ragged last token tile, N = 544 = 4 x 128 + 32);
            # t192_<G>: the same through the 192 x 192 persistent kernel (two-slot ring; N = 544 = 2 x 192 + 160, 1024 = 5 x 192 + 64)
            pers = force.startswith("persist") or force.startswith("t192")
            os.environ["PF_S3_TILE_NOW"] = "128" if pers else force
            os.environ["PF_S3_T192"] = "2" if force.startswith("t192") else "0"
            if pers:
                os.environ["PF_S3_PERSIST"], os.environ["PF_S3_GRID"] = "2", force.split("_")[1] if "_" in force else force[7:]
            else:
                os.environ["PF_S3_PERSIST"] = "0"
            y = torch.zeros(M, N, device=DEV)
            o.conv_split3(x3, pw3, y, act=act, res=r1, res2=r2)
            y3 = torch.zeros(3, M, N, dtype=torch.bfloat16, device=DEV)
            o.conv_split3(x3, pw3, y3, act=act, res=r1, res2=r2)
            e1 = float((y.double() - ref).abs().max()) / den
            e3 = float((y3.double().sum(0) - ref).abs().max()) / den
            errs += [e1, e3]
        for k in ("PF_S3_TILE_NOW", "PF_S3_PERSIST", "PF_S3_GRID", "PF_S3_T192"):
            os.environ.pop(k, None)
        yf = torch.zeros(1, 1, M, N, device=DEV)
        o.conv(x.view(1, 1, M, K), pw, yf, act=act, res=r1.view(1, 1, M, N) if res else None, res2=r2.view(1, 1, M, N) if res2 else None)
        ef = float((yf.view(M, N).double() - ref).abs().max()) / den
        info.append(f"K={K} N={N}: split {e1:.2e} (planes out {e3:.2e}) vs f32 kernel {ef:.2e}")
    # the fused producers: LayerNorm and attention outputs written as planes == the host split of their float32 outputs, bit for bit
    xs = torch.randn(2 * 1037, 1024, generator=g).to(DEV) * 3 + 0.5
    gam, bet = torch.randn(1024, generator=g).to(DEV), torch.randn(1024, generator=g).to(DEV)
    yf = torch.empty_like(xs)
    o.layernorm(xs, yf, gam, bet, 1e-6)
    y3 = torch.zeros(3, 2 * 1037, 1024, dtype=torch.bfloat16, device=DEV)
    o.layernorm_split3(xs, y3, gam, bet, 1e-6)
    ln_same = bool((y3.cpu() == torch.stack(pk.split3(yf.cpu()))).all())
    qkv = torch.randn(2 * 1037, 3 * 1024, generator=g).to(DEV)
    af = torch.empty(2 * 1037, 1024, device=DEV)
    o.vit_attention(qkv, af, 2, 1037, 16)
    a3 = torch.zeros(3, 2 * 1037, 1024, dtype=torch.bfloat16, device=DEV)
    o.vit_attention(qkv, a3, 2, 1037, 16)
    at_same = bool((a3.cpu() == torch.stack(pk.split3(af.cpu()))).all())
    info.append(f"layernorm planes identical={ln_same} attention planes identical={at_same}")
    if not (ln_same and at_same):
        return float("inf"), 2e-6, "; ".join(info)
    # CHUNK-MAJOR planes ([3, cols/32, rows, 32]: the layout the ViT blocks hand from producer to GEMM): every producer writes the same bits as
    # its row-major form, and a GEMM fed / storing chunk-major planes returns the same bits, through the one-tile and the persistent kernel
    M2 = 2 * 1037
    y3k = torch.zeros(3, 32, M2, 32, dtype=torch.bfloat16, device=DEV)
    o.layernorm_split3(xs, y3k, gam, bet, 1e-6)
    k_ln = bool((pk.kmajor_to_rows(y3k) == y3).all())
    a3k = torch.zeros(3, 32, M2, 32, dtype=torch.bfloat16, device=DEV)
    o.vit_attention(qkv, a3k, 2, 1037, 16)
    k_at = bool((pk.kmajor_to_rows(a3k) == a3).all())
    q3 = torch.stack(pk.split3(qkv.cpu())).contiguous().to(DEV)
    s3, s3k = torch.zeros_like(a3), torch.zeros_like(a3k)
    o.vit_attention(q3, s3, 2, 1037, 16)
    o.vit_attention(q3, s3k, 2, 1037, 16)
    k_as = bool((pk.kmajor_to_rows(s3k) == s3).all())
    w = torch.randn(4096, 1024, generator=g) / 32
    b = torch.randn(4096, generator=g)
    pr, pc = pk.pack_conv_split3(w, b, kmajor=False).to(DEV), pk.pack_conv_split3(w, b, kmajor=True).to(DEV)
    k_gemm = True
    for pers in ("0", "2"):
        os.environ["PF_S3_PERSIST"], os.environ["PF_S3_GRID"] = pers, "40"
        yr = torch.zeros(3, M2, 4096, dtype=torch.bfloat16, device=DEV)
        o.conv_split3(y3, pr, yr, act="gelu")
        yk = torch.zeros(3, 128, M2, 32, dtype=torch.bfloat16, device=DEV)
        o.conv_split3(y3k, pc, yk, act="gelu")
        ym = torch.zeros(3, M2, 4096, dtype=torch.bfloat16, device=DEV)
        o.conv_split3(y3k, pr, ym, act="gelu")                      # mixed: chunk-major x, row-major w and y
        k_gemm = k_gemm and bool((pk.kmajor_to_rows(yk) == yr).all()) and bool((ym == yr).all())
    for k in ("PF_S3_PERSIST", "PF_S3_GRID"):
        os.environ.pop(k, None)
    info.append(f"chunk-major: layernorm={k_ln} attention(f32 qkv)={k_at} attention(split)={k_as} gemm in/out={k_gemm}")
    if not (k_ln and k_at and k_as and k_gemm):
        return float("inf"), 2e-6, "; ".join(info)
    return max(errs), 2e-6, "; ".join(info)


def gemm_split3_persist(dt):
    """the persistent tile walk of the split-precision GEMM (csrc/gemm_split3.hip, round 4) on the launch it was built for -- the BATCHED
    transform-domain GEMM of a three-step Winograd layer (planes = transform points, plain float32 store): bit-identical to the one-tile-per-block
    kernel (same chunk order, same six terms per accumulator) for several grid sizes, odd and even chunk counts, a ragged last token tile and
    N = 544 (five channel tiles, the last with 32 live columns); and float32-grade against float64."""
    import os
    o = hip()
    g = torch.Generator().manual_seed(77)
    worst, info = 0.0, []
    for (P, T, cin, cout, grids) in ((5, 1000, 544, 544, ("8", "40", "256")), (3, 517, 96, 160, ("8", "24")), (36, 300, 128, 256, ("64", "256")),
                                     (2, 2100, 1024, 128, ("16",))):
        V = torch.randn(P, T, cin, generator=g)
        U = torch.randn(P, cout, cin, generator=g) / cin ** 0.5
        rows = -(-cout // 16) * 16
        Up = torch.zeros(P, rows, cin)
        Up[:, :cout] = U
        V3 = torch.stack(pk.split3(V)).contiguous().to(DEV)
        U3 = torch.stack(pk.split3(Up)).contiguous().to(DEV)
        ref = torch.einsum("ptk,pnk->ptn", V.double().to(DEV), U.double().to(DEV))
        den = max(1.0, float(ref.abs().max()))
        os.environ["PF_S3_TILE_NOW"], os.environ["PF_S3_PERSIST"] = "128", "0"
        y0 = torch.full((P, T, cout), float("nan"), device=DEV)
        o.gemm_planes_split3(V3, U3, y0, T, cin, cout)
        e0 = float((y0.double() - ref).abs().max()) / den
        same = True
        os.environ["PF_S3_T192"] = "0"
        for gr in grids:
            os.environ["PF_S3_PERSIST"], os.environ["PF_S3_GRID"] = "2", gr
            y1 = torch.full((P, T, cout), float("nan"), device=DEV)
            o.gemm_planes_split3(V3, U3, y1, T, cin, cout)
            same = same and bool((y1 == y0).all())
        # the 192 x 192 persistent kernel (two-slot ring): same chunk order and term order per accumulator -> the same bits
        os.environ["PF_S3_T192"] = "2"
        same192 = True
        for gr in grids:
            os.environ["PF_S3_PERSIST"], os.environ["PF_S3_GRID"] = "2", gr
            y1 = torch.full((P, T, cout), float("nan"), device=DEV)
            o.gemm_planes_split3(V3, U3, y1, T, cin, cout)
            same192 = same192 and bool((y1 == y0).all())
        # chunk-major operands ([plane][point][K/32][rows][32], what csrc/winograd.hip and PackedConv.wino_u3 hand to the kernel): same bits again,
        # through both kernels
        V3k = V3.view(3, P, T, cin // 32, 32).permute(0, 1, 3, 2, 4).contiguous()
        U3k = U3.view(3, P, rows, cin // 32, 32).permute(0, 1, 3, 2, 4).contiguous()
        for pers, gr, t192 in (("0", "0", "0"), ("2", grids[0], "0"), ("2", grids[-1], "0"), ("2", grids[0], "2"), ("2", grids[-1], "2")):
            os.environ["PF_S3_PERSIST"], os.environ["PF_S3_GRID"], os.environ["PF_S3_T192"] = pers, gr, t192
            if gr == "0":
                os.environ.pop("PF_S3_GRID")
            y1 = torch.full((P, T, cout), float("nan"), device=DEV)
            o.gemm_planes_split3(V3k, U3k, y1, T, cin, cout)
            if t192 == "2":
                same192 = same192 and bool((y1 == y0).all())
            else:
                same = same and bool((y1 == y0).all())
        for k in ("PF_S3_TILE_NOW", "PF_S3_PERSIST", "PF_S3_GRID", "PF_S3_T192"):
            os.environ.pop(k, None)
        info.append(f"P{P} T{T} {cin}->{cout}: err {e0:.2e} persistent==one-tile==chunk-major {same} 192-tile kernel identical {same192}")
        worst = max(worst, e0 if (same and same192) else float("inf"))
    return worst, 2e-6, "; ".join(info)


def conv1x1_split3(dt):
    """float32 1x1 convolution through the split-precision kernel that splits its activations in the loader (csrc/conv1x1_split3.hip, forced with
    PF_CONV1X1_SPLIT3=2) against float64 on the same float32 operands, beside the f32-MFMA kernel's own error (`_direct`): ragged token counts,
    1 .. 32 K chunks, channel counts off the 128-wide tile, channel-slice views on both sides, every epilogue option, ReLU on the input, and
    operands spanning twelve decades.  The split route must be float32-grade: no worse than 2e-6 of max|ref|."""
    import os
    o = hip()
    g = torch.Generator().manual_seed(171)
    os.environ["PF_CONV1X1_SPLIT3"] = "2"
    _switches_changed()
    worst, info = 0.0, []
    try:
        for (B, H, W, K, N, act, relu_in, res, res2, scale, xe, ye, wide) in (
                (1, 17, 61, 32, 128, None, False, False, False, False, 0, 0, False),          # one chunk, ragged tokens (1037)
                (2, 40, 52, 64, 80, "relu", False, True, False, False, 16, 8, False),         # two chunks, N below one tile, views
                (1, 37, 50, 96, 544, "gelu", True, True, True, True, 32, 16, False),          # three chunks, 544 = 4 x 128 + 32
                (3, 28, 37, 256, 128, "softplus", False, False, False, True, 0, 0, False),
                (1, 56, 74, 1024, 256, None, False, True, False, False, 0, 0, False),         # 32 chunks
                (1, 33, 47, 128, 36, None, False, False, False, False, 0, 8, True),           # N % 16 != 0; operands over twelve decades
                (1, 1, 5, 64, 64, "relu", True, False, False, False, 0, 0, False),             # five tokens: one partial tile, half a channel tile (zero-page weight rows)
                (1, 181, 182, 64, 64, None, False, True, False, False, 0, 0, False)):          # 32 942 tokens: the DEFAULT grid already walks two tiles per block
            w = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            sc = (0.5 + torch.rand(N, generator=g)) if scale else None
            pw = pk.pack_conv(w.view(N, K, 1, 1), b, dtype=torch.float32, scale=sc).to(DEV)
            assert pw.w3 is not None
            M = B * H * W
            xb = torch.randn(B, H, W, K + xe, generator=g)
            if wide:
                xb = xb * torch.logspace(-6, 6, K + xe)
            xb = xb.to(DEV)
            x = xb[..., xe // 2: xe // 2 + K]
            r1 = torch.randn(B, H, W, pw.cout, generator=g).to(DEV) if res else None
            r2 = torch.randn(B, H, W, pw.cout, generator=g).to(DEV) if res2 else None
            xd = torch.relu(x.double()) if relu_in else x.double()
            ref = xd.reshape(M, K) @ w.double().t().to(DEV) + b.double().to(DEV)
            if act == "gelu":
                ref = torch.nn.functional.gelu(ref)
            elif act == "relu":
                ref = torch.relu(ref)
            elif act == "softplus":
                ref = torch.nn.functional.softplus(ref)
            if sc is not None:
                ref = ref * sc.double().to(DEV)
            if r1 is not None:
                ref = ref + r1.double().reshape(M, -1)[:, :N]
            if r2 is not None:
                ref = ref + r2.double().reshape(M, -1)[:, :N]
            den = max(1.0, float(ref.abs().max()))
            es, first = [], None
            # (None, slots): the split kernel with PF_C1_SLOTS token-tile slots per channel tile -- default grid, then 3 and 1 slots (every block walks
            # many tiles: the persistent chunk stream across tile boundaries, ragged last tile) -- all bit-identical; (True, None): the f32-MFMA kernel
            for direct, slots in ((None, None), (None, "3"), (None, "1"), (True, None)):
                if slots is None:
                    os.environ.pop("PF_C1_SLOTS", None)
                else:
                    os.environ["PF_C1_SLOTS"] = slots
                yb = torch.full((B, H, W, pw.cout + ye), -7.0, device=DEV)
                y = yb[..., ye // 2: ye // 2 + pw.cout]
                _flush_caches()      # cold operands: a hand-counted wait that leaves a needed piece in flight shows on the FIRST touch, not on warm re-runs
                o.conv(x, pw, y, act=act, relu_in=relu_in, res=r1, res2=r2, _direct=direct)
                untouched = bool((yb[..., :ye // 2] == -7.0).all() and (yb[..., ye // 2 + pw.cout:] == -7.0).all())
                e = float((y.reshape(M, -1)[:, :N].double() - ref).abs().max()) / den
                if direct is None:
                    if first is None:
                        first = yb.clone()
                    elif not bool((yb == first).all()):
                        e = float("inf")                 # the tile walk changed the numbers
                if slots is None:
                    es.append(e if untouched and torch.isfinite(y).all() else float("inf"))
                elif not (untouched and e < float("inf")):
                    es[0] = float("inf")
            os.environ.pop("PF_C1_SLOTS", None)
            from patchfusion_amd import hip_ops
            route = hip_ops.HipOps._conv_plan(x, pw, y, 1, 0, act, relu_in, r1, r2, None)[0]
            info.append(f"M={M} {K}->{N}: split {es[0]:.2e} f32 kernel {es[1]:.2e} route {route}")
            worst = max(worst, es[0] if route == "s3_1x1" else float("inf"))
    finally:
        os.environ.pop("PF_CONV1X1_SPLIT3", None)
        _switches_changed()
    torch.cuda.synchronize()
    return worst, 2e-6, "; ".join(info)


def swin_ops(dt):
    errs = []
    for (B, H, W, C, heads) in ((1, 14, 19, 64, 32), (1, 28, 37, 64, 16), (2, 30, 25, 32, 8), (1, 13, 24, 256, 8), (1, 12, 12, 128, 8), (1, 17, 12, 256, 16), (1, 12, 24, 64, 8)):
        for shift in (0, 6):
            Hp, Wp = (H + 11) // 12 * 12, (W + 11) // 12 * 12
            x = _rand((B, H, W, C), dt, H * W + shift)
            g, b = 1 + 0.1 * _rand((C,), torch.float32, 1), 0.1 * _rand((C,), torch.float32, 2)
            nt = B * Hp * Wp
            qkv = _rand((nt, 3 * C), dt, 5)
            bt = _rand((529, heads), torch.float32, 6, 0.5)
            pr = _rand((nt, C), dt, 8)
            pos = _rand((H * W, C), torch.float32, 9)
            res = []
            for o in (hip(), ref_ops):
                xw = torch.zeros((nt, C), dtype=dt, device=DEV)
                o.swin_ln_partition(x, xw, g, b, 1e-5, shift)
                ao = torch.zeros((nt, C), dtype=dt, device=DEV)
                o.swin_window_attention(qkv, ao, bt, B, Hp, Wp, C, heads, shift)
                y = torch.zeros((B, H, W, C), dtype=dt, device=DEV)
                o.swin_unpartition_add(pr, x, y, shift)
                x2 = x.clone().view(B, H * W, C)
                o.add_rowwise(x2, pos)
                res.append((xw, ao, y, x2))
            errs += [_err(a, c) for a, c in zip(res[0], res[1])]
    torch.cuda.synchronize()
    return max(errs), _tol(dt, 1e-4, 2e-2), "swin ops"


def resize_ops(dt):
    errs = []
    # (the last case: W + OW above the ~7.4k the source-aligned kernel's LDS tables hold -> the library falls through to the output-walking kernel)
    for (h, w, oh, ow, C) in ((14, 19, 28, 37, 64), (49, 64, 56, 74, 32), (224, 296, 392, 518, 8), (12, 16, 14, 19, 128), (8, 11, 8, 11, 64), (2, 4000, 3, 4100, 8)):
        x = _rand((2, h, w, C), dt, h)
        add = _rand((2, oh, ow, C), dt, w)
        res = []
        for o in (hip(), ref_ops):
            buf = torch.zeros((2, oh, ow, C + 16), dtype=dt, device=DEV)
            o.resize(x, buf[..., 8:8 + C])
            y2 = torch.zeros((2, oh, ow, C), dtype=dt, device=DEV)
            o.resize(x, y2, add=add)
            res.append((buf, y2))
        errs += [_err(a, c) for a, c in zip(res[0], res[1])]
    # the source-aligned kernels (version 2, default) against the output-walking ones (PF_RESIZE_V2=0): every output pixel written exactly
    # once, bit-identical -- up-sampling, down-sampling, identity, a single output row / column
    import os
    same = True
    for (h, w, oh, ow, C) in ((14, 19, 28, 37, 64), (224, 296, 392, 518, 8), (56, 74, 28, 37, 32), (37, 50, 9, 200, 16), (8, 11, 8, 11, 64),
                              (5, 7, 1, 1, 8), (1, 1, 6, 9, 8), (30, 3, 31, 2, 24)):
        x = _rand((2, h, w, C), dt, h + 100)
        add = _rand((2, oh, ow, C), dt, w + 100)
        outs = []
        for v2 in ("1", "0"):
            os.environ["PF_RESIZE_V2"] = v2
            y1 = torch.full((2, oh, ow, C + 16), -7.0, dtype=dt, device=DEV)
            hip().resize(x, y1[..., 8:8 + C])
            y2 = torch.full((2, oh, ow, C), -7.0, dtype=dt, device=DEV)
            hip().resize(x, y2, add=add)
            y3 = torch.full((2, oh, ow, 3 * C), -7.0, dtype=dt, device=DEV)
            hip().resize_concat([x, add, x], y3)
            outs.append((y1, y2, y3))
        os.environ.pop("PF_RESIZE_V2", None)
        same = same and all(bool((a == b).all()) for a, b in zip(*outs))
    if not same:
        return float("inf"), _tol(dt, 1e-4, 1e-2), "resize version 2 differs from the output-walking kernels"
    # the Upv1 concat buffer in one launch: 3 sources of different sizes / 2 sources into a channel slice of a wider buffer
    e0, t0, g0 = _rand((2, 49, 64, 64), dt, 31), _rand((2, 28, 37, 128), dt, 32), _rand((2, 28, 37, 128), dt, 33)
    res = []
    for o in (hip(), ref_ops):
        u = torch.zeros((2, 56, 74, 320), dtype=dt, device=DEV)
        o.resize_concat([e0, t0, g0], u)
        u5 = torch.zeros((2, 56, 74, 32 + 256), dtype=dt, device=DEV)
        o.resize_concat([t0, g0], u5[..., 32:])
        res.append((u, u5))
    errs += [_err(a, c) for a, c in zip(res[0], res[1])]
    # f32 source -> dt destination (embedding upsample into the CLB buffer) and planar helpers
    img = torch.rand(3, 96, 130, generator=torch.Generator().manual_seed(5)).to(DEV)
    boxes = torch.tensor([[0, 0, 65, 48], [65, 48, 130, 96], [13, 7, 78, 55]], dtype=torch.int32, device=DEV)
    res = []
    for o in (hip(), ref_ops):
        out = torch.zeros((3, 3, 28, 42), dtype=torch.float32, device=DEV)
        o.crop_resize(img, boxes, out)
        a = torch.rand(40, 52, generator=torch.Generator().manual_seed(6)).to(DEV)
        n, bl = torch.zeros((96, 130), device=DEV), torch.zeros((96, 130), device=DEV)
        o.resize_nearest_f32(a, n)
        o.resize_bilinear_f32(a, bl)
        res.append((out, n, bl))
    errs += [_err(a, c) for a, c in zip(res[0], res[1])]
    torch.cuda.synchronize()
    return max(errs), _tol(dt, 1e-4, 1e-2), "resize/crop"  # bilinear source-coordinate rounding (scale*dst in f32)


def roi_ops(dt):
    errs = []
    ph = 112
    feats = [(4, 6, 64), (28, 37, 64), (112, 154, 32)]
    rois = torch.tensor([[0, 0.0, 0.0, 77.0, 56.0], [0, 77.0, 56.0, 154.0, 112.0], [0, 38.5, 28.0, 115.5, 84.0],
                         [0, 100.25, 60.5, 177.25, 116.5], [0, -90.0, -70.0, -13.0, -14.0]], device=DEV)
    for (h, w, C) in feats:
        f = _rand((1, h, w, C), dt, h)
        res = []
        for o in (hip(), ref_ops):
            y = torch.zeros((5, h, w, 2 * C), dtype=dt, device=DEV)
            o.roi_align(f, rois, y[..., C:], h / ph)
            res.append(y)
        errs.append(_err(res[0], res[1]))
    # multi-sample bins (roi larger than the output grid) and the planar depth variant
    f = _rand((2, 16, 16, 8), dt, 3)
    r2 = torch.tensor([[1, 0.0, 0.0, 16.0, 16.0], [0, 2.0, 3.0, 14.0, 12.0]], device=DEV)
    d = torch.rand(1, 1, 112, 154, generator=torch.Generator().manual_seed(9)).to(DEV)
    res = []
    for o in (hip(), ref_ops):
        y = torch.zeros((2, 4, 5, 8), dtype=dt, device=DEV)
        o.roi_align(f, r2, y, 1.0)
        yd = torch.zeros((5, 1, 112, 154), device=DEV)
        o.roi_align_depth(d, rois, yd, 1.0)
        res.append((y, yd))
    errs += [_err(a, c) for a, c in zip(res[0], res[1])]
    torch.cuda.synchronize()
    return max(errs), _tol(dt, 1e-5, 1e-2), "roi_align"


def misc_ops(dt):
    errs = []
    x = _rand((2, 49, 65, 32), dt, 1)
    cd, fd = torch.rand(2, 1, 20, 30).to(DEV), torch.rand(2, 20, 30).to(DEV)
    crops = torch.rand(2, 3, 20, 30).to(DEV)
    res = []
    for o in (hip(), ref_ops):
        y = torch.zeros((2, 24, 32, 32), dtype=dt, device=DEV)
        o.maxpool2(x, y)
        c = torch.zeros((2, 49, 65, 64), dtype=dt, device=DEV)
        o.copy_channels(x, c[..., 32:])
        p = torch.zeros((2, 20, 30, 8), dtype=dt, device=DEV)
        o.pack_fusion_input(cd, fd, crops, p)
        n = o.nhwc_to_nchw(c[..., 32:])
        res.append((y, c, p, n))
    errs += [_err(a, c) for a, c in zip(res[0], res[1])]
    torch.cuda.synchronize()
    return max(errs), _tol(dt, 1e-6, 1e-2), "maxpool/copy/pack/nchw"


def bins_ops(dt):
    errs = []
    for n_attr, (hp, wp, h, w) in ((16, (4, 6, 8, 11)), (8, (8, 11, 16, 22)), (1, (32, 44, 64, 88))):
        A = torch.nn.functional.softplus(_rand((2, h, w, (n_attr + 3) // 4 * 4), torch.float32, n_attr))
        bp = torch.nn.functional.softplus(_rand((2, hp, wp, 64), torch.float32, 3))
        res = []
        for o in (hip(), ref_ops):
            out = torch.zeros((2, h, w, 64), device=DEV)
            o.attractor(A, n_attr, bp, out)
            res.append(out)
        errs.append(_err(res[0], res[1]))
    # the bounded layers and the other attractor type / kind (bin_centers_type 'normed' / 'hybrid*', attractor.py:60-136)
    A2 = torch.relu(_rand((2, 16, 22, 32), torch.float32, 21))
    bp = torch.rand((2, 8, 11, 64), generator=torch.Generator().manual_seed(22)).to(DEV)
    for kw in (dict(a_stride=2, a_eps=1e-3), dict(a_stride=2, a_eps=1e-3, attractor_type="exp", kind="sum"), dict(attractor_type="exp"), dict(kind="sum")):
        res = []
        for o in (hip(), ref_ops):
            out = torch.zeros((2, 16, 22, 64), device=DEV)
            o.attractor(A2, 16, bp, out, **kw)
            res.append(out)
        errs.append(_err(res[0], res[1]))
    x = torch.relu(_rand((2, 9, 13, 64), torch.float32, 23))
    for bounded, normalize in ((True, True), (True, False), (False, True)):
        res = [o.seed_bin_centers(x, torch.zeros((2, 9, 13, 64), device=DEV), 1e-3, 80.0, bounded, normalize) for o in (hip(), ref_ops)]
        errs.append(_err(res[0], res[1]) / (1.0 if normalize else 80.0))
    for nb in (64, 16):
        b = _rand((3, 17, 19, nb), torch.float32, 24) * 0.6 + 0.5          # some outside [0, 1] -> clipped, unsorted, with ties
        b[0, :4] = b[0, :4].round()
        res = [o.bounded_bin_centers(b, torch.zeros_like(b), 1e-3, 80.0) for o in (hip(), ref_ops)]
        errs.append(float((res[0] - res[1]).abs().max()) / 80.0)
        assert bool((res[0][..., 1:] >= res[0][..., :-1]).all())
    pt = torch.nn.functional.softplus(_rand((2, 56, 77, 4), torch.float32, 5) + torch.tensor([0, 0, -4.0, 1.0], device=DEV))
    cen = torch.nn.functional.softplus(_rand((2, 32, 44, 64), torch.float32, 6))
    res = []
    for o in (hip(), ref_ops):
        d = torch.zeros((2, 56, 77), device=DEV)
        o.logbinom_depth(pt, cen, d, 0.0212, 50.0)
        res.append(d)
    errs.append(_err(res[0], res[1]))
    torch.cuda.synchronize()
    return max(errs), 1e-4, "attractor/logbinom"


def bins_tail(dt):
    """the fused metric-bins tail (pf_bins_tail) against the four launches it replaces (HIP kernels) and against the torch reference of
    those (tests/fake_ops.py): with and without the rel channels, a pixel count that is not a multiple of the 16-pixel wave step"""
    o = hip()
    g = torch.Generator().manual_seed(91)
    errs, info = [], []
    for ctot in (168, 160):
        B, H, W, he, we = 2, 37, 50, 21, 29
        w0 = torch.randn(80, ctot, 1, 1, generator=g) / ctot ** 0.5
        w2 = torch.randn(4, 80, 1, 1, generator=g) / 80 ** 0.5
        mlp0 = pk.pack_conv(w0, torch.randn(80, generator=g) * 0.1, dtype=torch.float32).to(DEV)
        mlp2 = pk.pack_conv(w2, torch.randn(4, generator=g) * 0.1, dtype=torch.float32).to(DEV)
        tw = pk.bins_tail_weights(mlp0, mlp2, 128)
        assert tw is not None and tw.nq == (11 if ctot == 168 else 10)
        tw = tw.to(DEV)
        clb = torch.randn(B, H, W, ctot, generator=g).to(DEV)
        clb[..., 32:160] = 123.0                                  # the embedding slice of the buffer must NOT be read by the fused kernel
        emb = torch.randn(B, he, we, 128, generator=g).to(DEV)
        cen = (torch.rand(B, he, we, 64, generator=g) * 5 + 0.5).sort(-1).values.contiguous().to(DEV)
        d_f = torch.zeros(B, H, W, device=DEV)
        o.bins_tail(clb, emb, tw, cen, d_f, 0.0212, 50.0)
        outs = []
        for oo in (o, ref_ops):
            buf = clb.clone()
            oo.resize(emb, buf[..., 32:160])
            t = torch.zeros(B, H, W, 80, device=DEV)
            oo.conv(buf, mlp0, t, act="gelu")
            pt = torch.zeros(B, H, W, 4, device=DEV)
            oo.conv(t, mlp2, pt, act="softplus")
            d = torch.zeros(B, H, W, device=DEV)
            oo.logbinom_depth(pt, cen, d, 0.0212, 50.0)
            outs.append(d)
        e_hip, e_ref = _err(d_f, outs[0]), _err(d_f, outs[1])
        errs += [e_hip, e_ref]
        info.append(f"ctot {ctot}: vs four HIP launches {e_hip:.2e}, vs torch {e_ref:.2e}")
    return max(errs), 1e-4, "; ".join(info)


def stitch_ops(dt):
    ph, pw = 28, 42
    depth = torch.rand(6, ph, pw, generator=torch.Generator().manual_seed(1)).to(DEV) + 0.5
    mask = torch.rand(ph, pw, generator=torch.Generator().manual_seed(2)).to(DEV) + 1e-3
    yx = torch.tensor([[0, 0], [0, 42], [28, 0], [28, 42]], dtype=torch.int32, device=DEV)
    rawmask = torch.rand(40, 60, generator=torch.Generator().manual_seed(3)).to(DEV) + 1e-3
    res = []
    for o in (hip(), ref_ops):
        pred, cnt, avg = torch.zeros(56, 84, device=DEV), torch.zeros(56, 84, device=DEV), torch.zeros(56, 84, device=DEV)
        o.stitch_init(pred, cnt, depth[:4].contiguous(), mask, yx)
        o.stitch_finish_init(avg, pred, cnt)
        o.stitch_update(avg, cnt, depth[4], mask, 14, 21)
        a2, c2 = torch.zeros(80, 120, device=DEV), torch.zeros(80, 120, device=DEV)
        o.resize_nearest_f32(avg, a2)
        o.resize_bilinear_f32(cnt, c2)
        o.stitch_update(a2, c2, depth[5], rawmask, 33, 47)
        res.append((avg, cnt, a2, c2))
    errs = [_err(a, c) for a, c in zip(res[0], res[1])]
    torch.cuda.synchronize()
    return max(errs), 1e-5, "stitch"


CHECKS = {
    "conv_gemm_qkv": conv_gemm_qkv, "conv_gemm_fc2_inplace": conv_gemm_fc2_inplace, "conv_gemm_gelu": conv_gemm_gelu,
    "conv3x3_rcu": conv3x3_rcu, "conv3x3_stride2": conv3x3_stride2, "conv3x3_small_cin": conv3x3_small_cin,
    "conv3x3_n160_views": conv3x3_n160_views, "conv3x3_n544": conv3x3_n544, "conv1x1_cout1_f32out": conv1x1_cout1_f32out,
    "conv1x1_cout80": conv1x1_cout80, "conv1x1_cout16": conv1x1_cout16, "conv3x3_nobias_48": conv3x3_nobias_48,
    "conv_big_3x3_rcu": conv_big_3x3_rcu, "conv_big_3x3_n544_views": conv_big_3x3_n544_views,
    "conv_big_3x3_n160": conv_big_3x3_n160, "conv_big_stride2": conv_big_stride2,
    "conv_halo_n32": conv_halo_n32, "conv_halo_n64_res": conv_halo_n64_res, "conv_halo_n32_cin544": conv_halo_n32_cin544,
    "conv_halo_bn192_n544": conv_halo_bn192_n544, "conv_halo_bn128_n224": conv_halo_bn128_n224, "conv_halo_bn192_n160": conv_halo_bn192_n160,
    "conv_big_gemm_scale_inplace": conv_big_gemm_scale_inplace, "conv_big_gemm_gelu_k1024": conv_big_gemm_gelu_k1024,
    "conv_big_transpose": conv_big_transpose,
    "conv_split_n272_res_views": conv_split_n272_res_views, "conv_split_gemm_n544_inplace": conv_split_gemm_n544_inplace,
    "conv_gemm_vitl_linear_shape": conv_gemm_vitl_linear_shape, "conv_bf16_pp": conv_bf16_pp, "conv_dominant_launch": conv_dominant_launch,
    "conv_transpose": conv_transpose, "patch_embed_tokens": patch_embed_tokens, "layernorm": layernorm,
    "vit_attention": vit_attention, "vit_attention_split": vit_attention_split, "gemm_split3": gemm_split3, "gemm_split3_persist": gemm_split3_persist, "conv1x1_split3": conv1x1_split3, "vit_attention_split3": vit_attention_split3, "vit_attention_split3_v2": vit_attention_split3_v2, "swin_ops": swin_ops, "resize_ops": resize_ops, "roi_ops": roi_ops,
    "conv_winograd": conv_winograd, "conv_winograd_subbatch": conv_winograd_subbatch, "conv_winograd_fused": conv_winograd_fused, "misc_ops": misc_ops, "bins_ops": bins_ops, "bins_tail": bins_tail, "stitch_ops": stitch_ops,
}
F32_ONLY = {"conv1x1_split3", "bins_ops", "bins_tail", "stitch_ops", "conv_winograd", "conv_winograd_subbatch", "conv_winograd_fused", "vit_attention_split", "gemm_split3", "gemm_split3_persist", "vit_attention_split3", "vit_attention_split3_v2"}
DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16}
