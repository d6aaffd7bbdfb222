"""CPU tests of the host-side dispatch rules of round 3 (no kernel is launched): which float32 3x3 layers get which Winograd filters at packing
time, and which form hip_ops picks per call (fused kernel / three steps with the split or f32 GEMM / direct kernel)."""
import importlib
import os

import pytest
import torch

from patchfusion_amd import packing as pk


def _pack(cout, cin, k=3, **kw):
    return pk.pack_conv(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5, torch.zeros(cout), dtype=torch.float32, **kw)


def test_packing_chooses_winograd_filters_by_layer_shape(monkeypatch):
    monkeypatch.delenv("PF_WINO_SPLIT3", raising=False)
    monkeypatch.delenv("PF_WINO_FUSED_SMALL", raising=False)
    big = _pack(544, 544)                  # three-step eligible: all three filter sets
    assert big.wino_m == 4 and big.wino_u is not None and big.wino_up is not None and big.wino_u3 is not None
    P, rows, K = big.wino_u.shape
    assert tuple(big.wino_u3.shape) == (3, P, K // 32, rows, 32) and big.wino_u3.dtype == torch.bfloat16      # chunk-major planes
    u3_rows = big.wino_u3.permute(0, 1, 3, 2, 4).reshape(3, P, rows, K)
    assert bool((u3_rows.double().sum(0) == big.wino_u.double()).all())          # the planes are an exact split of G g G^T
    small = _pack(32, 64)                  # below the three-step channel threshold: fused filters only
    assert small.wino_m == 4 and small.wino_u is None and small.wino_u3 is None and small.wino_up is not None
    assert _pack(32, 8).wino_m == 0 and _pack(32, 8).wino_up is None                  # Cin < 32: direct kernel
    assert _pack(64, 64, k=1).wino_m == 0                                             # not a 3x3 layer
    assert pk.pack_conv(torch.randn(64, 64, 3, 3), torch.zeros(64), dtype=torch.bfloat16).wino_m == 0
    assert pk.pack_conv(torch.randn(64, 64, 3, 3), torch.zeros(64), dtype=torch.float32, scale=torch.ones(64)).wino_m == 0   # LayerScale epilogue
    monkeypatch.setenv("PF_WINO_FUSED_SMALL", "0")
    assert _pack(32, 64).wino_m == 0
    monkeypatch.setenv("PF_WINO_SPLIT3", "0")
    assert _pack(544, 544).wino_u3 is None
    # applies(): the call-time half
    assert pk.winograd_applies(small, 10 ** 6, 1, 1, "relu") and not pk.winograd_applies(small, 10 ** 6, 2, 1, None)
    assert not pk.winograd_applies(small, 10 ** 6, 1, 1, "gelu") and not pk.winograd_applies(_pack(32, 8), 10 ** 6, 1, 1, None)


def test_fused_versus_three_step_rule(monkeypatch):
    try:
        hip_ops = importlib.import_module("patchfusion_amd.hip_ops")
    except Exception as e:                 # the product path needs libpf_hip.so even to import (no CPU fallback)
        pytest.skip(f"libpf_hip.so not built: {e}")
    for v in ("PF_WINO_FUSED", "PF_WINO_SPLIT3", "PF_WINO_FUSED_SMALL"):
        monkeypatch.delenv(v, raising=False)
    p544, p768, p768_256, p64_32 = _pack(544, 544), _pack(768, 768), _pack(256, 768), _pack(32, 64)

    def fw(*a):                      # (the switches are resolved once and cached: a test that flips them says so)
        hip_ops.refresh_env()
        return hip_ops._fused_wanted(*a)
    assert not fw(8, 392, 518, p544) and not fw(8, 224, 296, p768)          # >= 256 output channels (round 4): three steps with the split GEMM
    assert not fw(8, 224, 296, p768_256)
    assert fw(8, 224, 296, _pack(128, 256)) and fw(8, 392, 518, p64_32)     # fewer: fused kernel
    assert not fw(1, 28, 37, p768_256) and not fw(1, 56, 74, p64_32)        # too few blocks: three-step / direct
    monkeypatch.setenv("PF_WINO_SPLIT3", "0")                               # f32 GEMM: the old rule (only 768+ -> 768+ stays three-step)
    assert fw(8, 392, 518, p544) and not fw(8, 224, 296, p768) and fw(8, 224, 296, p768_256)
    monkeypatch.setenv("PF_WINO_FUSED", "0")
    assert not fw(8, 392, 518, p544) and not fw(8, 392, 518, p64_32)
    monkeypatch.setenv("PF_WINO_FUSED", "2")
    assert fw(1, 28, 37, p768)


def test_three_step_layers_walk_their_tiles_in_capped_windows(monkeypatch):
    """hip_ops.wino3_window (round 5): the V + M arena pair of a three-step layer is capped (PF_WS_CAP_GB, default 2.5); larger layers run in windows of
    their Winograd tiles -- multiples of the GEMM's 192-token tile, the last one ragged -- through that one pair."""
    try:
        hip_ops = importlib.import_module("patchfusion_amd.hip_ops")
    except Exception as e:
        pytest.skip(f"libpf_hip.so not built: {e}")
    from types import SimpleNamespace as NS
    monkeypatch.delenv("PF_WS_CAP_GB", raising=False)
    hip_ops.refresh_env()
    pw = NS(cin=544, cout=544)
    T = 8 * 98 * 130
    win, nwin, nV, nM = hip_ops.wino3_window(8, 392, 518, pw)
    assert win % 192 == 0 and nwin == -(-T // win) and (nV + nM) * 4 <= 2.5 * 2 ** 30 < (36 * (win + 192) * 544 * 10)
    assert nV * 4 >= 36 * win * 544 * 6 and nM == 36 * win * 544 and nwin == 8                      # the headline layer: eight windows of 2.3 GB
    small = hip_ops.wino3_window(8, 56, 74, NS(cin=256, cout=256))
    assert small[1] == 1 and small[0] == -(-(8 * 14 * 19) // 8) * 8                                 # fits: one window = all tiles (whole octets)
    monkeypatch.setenv("PF_WS_CAP_GB", "0.000001")                                                  # never below one GEMM tile of tokens
    hip_ops.refresh_env()
    assert hip_ops.wino3_window(8, 392, 518, pw)[:2] == (192, -(-T // 192))
    monkeypatch.setenv("PF_WS_CAP_GB", "100")
    hip_ops.refresh_env()
    assert hip_ops.wino3_window(8, 392, 518, pw)[:2] == (T, 1)
    monkeypatch.delenv("PF_WS_CAP_GB")
    hip_ops.refresh_env()


def test_split_gemm_kernel_choice_by_shape(monkeypatch):
    """pf_gemm_split3_route (csrc/gemm_split3.hip split3_route, no launch, no GPU): which of the four split-GEMM kernels a call runs on a 256-CU chip --
    the launches of the 4K ViT-L pass named in DESIGN.md 4g.  Round 4's rule: persistent from two rounds of 128 x 128 tiles on, 192 x 192 tiles when
    rounds of tiles per CU x tile cost (2.1 vs 1.0) is lower."""
    import ctypes as C
    try:
        from patchfusion_amd import _lib
        L = _lib.load()
    except Exception as e:
        pytest.skip(f"libpf_hip.so not built: {e}")
    for v in ("PF_S3_TILE_NOW", "PF_S3_PERSIST", "PF_S3_T192", "PF_S3_GRID"):
        monkeypatch.delenv(v, raising=False)
    T64, T128, P128, P192 = 0, 1, 2, 3

    def route(M, K, N, planes=1, cus=256):
        p = _lib.ConvParams()
        p.B, p.OH, p.OW, p.H, p.W, p.Cin, p.Cout, p.batch = 1, 1, M, 1, M, K, N, planes
        return L.pf_gemm_split3_route(C.byref(p), cus)

    T = 8 * 98 * 130
    assert route(T, 544, 544, 36) == P192                        # the dominant launch: 576 instead of 640 columns
    assert route(8 * 56 * 74, 768, 768, 36) == P192              # 768 = 4 x 192 = 6 x 128: the faster tile
    assert route(8 * 56 * 74, 768, 256, 36) == P128              # 256 would pad to 384
    assert route(8 * 56 * 74, 256, 256, 36) == P128
    assert route(8 * 1037, 1024, 3072) == P192 and route(8 * 1037, 1024, 4096) == P192        # qkv, fc1
    assert route(8 * 1037, 1024, 1024) == P128 and route(8 * 1037, 4096, 1024) == P128        # proj, fc2: 264 192-tiles on 256 CUs = two rounds
    assert route(1037, 1024, 4096) == T128 and route(1037, 1024, 1024) == T64                 # the coarse branch: one round of tiles or less
    assert route(8 * 1037, 64, 3072) == T128                     # K below three chunks: no stream to keep going
    assert route(8 * 14 * 19, 768, 768, 36) == P128 and route(8 * 28 * 37, 768, 768, 36) == P192 and route(70, 768, 768, 36) == T64        # small Winograd layers count the tiles of all 36 points
    monkeypatch.setenv("PF_S3_T192", "0")
    assert route(T, 544, 544, 36) == P128
    monkeypatch.setenv("PF_S3_T192", "2")
    assert route(8 * 1037, 1024, 1024) == P192
    monkeypatch.setenv("PF_S3_PERSIST", "0")
    assert route(T, 544, 544, 36) == T128
    assert L.pf_gemm_split3_route(None, 256) == -1


def test_float32_1x1_layers_carry_split_planes_and_route_by_shape(monkeypatch):
    """round 6: pack_conv gives float32 1x1 layers with Cin % 32 == 0 the three bf16 planes of their weight (chunk-major, an exact split), and hip_ops
    sends a call through csrc/conv1x1_split3.hip only where that kernel was measured to win (profiles/r6_conv1x1_split3.md)"""
    monkeypatch.delenv("PF_CONV1X1_SPLIT3", raising=False)
    w = torch.randn(80, 96, 1, 1) * torch.logspace(-3, 3, 96).view(1, 96, 1, 1)
    bn = (torch.rand(80) + 0.5, torch.randn(80), torch.randn(80), torch.rand(80) + 0.5)
    pw = pk.pack_conv(w, torch.randn(80), dtype=torch.float32, bn=bn)
    assert pw.w3 is not None and pw.w3.dtype == torch.bfloat16 and tuple(pw.w3.shape) == (3, 3, 80, 32)
    rows = pk.kmajor_to_rows(pw.w3)                                   # [3, rows, K]
    assert bool((rows.double().sum(0) == pw.w[:, :96].double()).all())   # planes == the packed (BatchNorm-folded) float32 weight, exactly
    assert _pack(64, 64).w3 is None and _pack(64, 40, k=1).w3 is None    # 3x3 layers and Cin % 32 != 0: no planes
    assert pk.pack_conv(torch.randn(64, 64, 1, 1), None, dtype=torch.bfloat16).w3 is None
    monkeypatch.setenv("PF_CONV1X1_SPLIT3", "0")
    assert _pack(64, 64, k=1).w3 is None
    monkeypatch.delenv("PF_CONV1X1_SPLIT3")
    try:
        hip_ops = importlib.import_module("patchfusion_amd.hip_ops")
    except Exception as e:
        pytest.skip(f"libpf_hip.so not built: {e}")

    def want(M, cout, cin):
        hip_ops.refresh_env()
        return hip_ops._conv1x1_split3_wanted(M, _pack(cout, cin, k=1))
    assert want(8 * 224 * 296, 128, 256) and want(66304, 1024, 256) and want(8 * 28 * 37, 1024, 1024) and want(33152, 128, 256)
    assert not want(1037, 1024, 1024)              # the coarse branch's token count: less than two tiles per CU
    assert want(203056, 96, 32) and want(203056, 128, 32)          # the 32 -> 96 / 128 linears of the first G2L level
    assert not want(203056, 32, 128) and not want(8 * 392 * 518, 32, 32) and not want(8 * 224 * 296, 4, 128)     # fewer than 64 output channels
    monkeypatch.setenv("PF_CONV1X1_SPLIT3", "2")
    assert want(1037, 32, 32)
    monkeypatch.setenv("PF_CONV1X1_SPLIT3", "0")
    assert not want(8 * 224 * 296, 128, 256)
    monkeypatch.delenv("PF_CONV1X1_SPLIT3")
    hip_ops.refresh_env()
