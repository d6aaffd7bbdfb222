"""CPU tests of the host side of the round-3 kernels: the three-way bf16 split and the split-GEMM weight planes (csrc/gemm_split3.hip),
and the fragment order of the fused metric-bins tail weights (csrc/imageops.hip bins_tail_kernel) replayed with the MFMA's lane mapping."""
import numpy as np
import torch

from patchfusion_amd import packing as pk
from tests.wino_fused_model import mfma_16x16x4


def test_split3_is_exact_over_the_activation_range():
    """exact for |x| < bf16 max (3.39e38: above it the leading plane rounds to inf) and down to ~1e-30 (below, the third plane leaves the
    bf16 normal range); activations and weights are O(1e-6 .. 1e4)"""
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * torch.logspace(-20, 20, 4096), torch.tensor([0.0, -0.0, 1.0, -1.0, 3.3e38, 1e-30, 1 + 2 ** -23, 1 - 2 ** -24])])
    h, m, l = pk.split3(x)
    assert h.dtype == m.dtype == l.dtype == torch.bfloat16
    assert bool((h.double() + m.double() + l.double() == x.double()).all())
    # the planes shrink by 2^-8 each (round to nearest): |m| <= 2^-8 |h|-ish, |l| <= 2^-16
    nz = x != 0
    assert float((m.float().abs()[nz] / x.abs()[nz]).max()) <= 2 ** -8
    assert float((l.float().abs()[nz] / x.abs()[nz]).max()) <= 2 ** -16


def test_pack_conv_split3_layout_and_six_term_product():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(10, 64, generator=g)
    b = torch.randn(10, generator=g)
    sc = torch.rand(10, generator=g)
    pw = pk.pack_conv_split3(w, b, scale=sc, kmajor=False)
    assert tuple(pw.w.shape) == (3, 16, 64) and pw.cout == 12 and pw.cout_real == 10 and pw.cin == 64
    assert bool((pw.w.double().sum(0)[:10] == w.double()).all()) and float(pw.w.float()[:, 10:].abs().max()) == 0.0
    # chunk-major planes (the default): [3, K/32, rows, 32], the same bits -- every 32-deep K chunk of all rows is one slab
    pk_ = pk.pack_conv_split3(w, b, scale=sc, kmajor=True)
    assert tuple(pk_.w.shape) == (3, 2, 16, 32) and pk_.w.is_contiguous() and bool((pk.kmajor_to_rows(pk_.w) == pw.w).all())
    assert bool((pk_.w[:, 1, 3] == pw.w[:, 3, 32:]).all()) and bool((pk.rows_to_kmajor(pw.w) == pk_.w).all())
    # the six products the kernel evaluates, in float64: error O(2^-24) of sum |x||w|, the three dropped terms
    x = torch.randn(33, 64, generator=g)
    xs, ws = torch.stack(pk.split3(x)).double(), pw.w.double()[:, :10]
    six = sum(xs[px] @ ws[pwi].t() for pwi, px in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)))
    exact = x.double() @ w.double().t()
    bound = (x.abs().double() @ w.abs().double().t()) * 3 * 2.0 ** -24 * 1.01      # |x_m w_l| + |x_l w_m| + |x_l w_l| <= 3 * 2^-24 |x||w|
    assert bool(((six - exact).abs() <= bound).all())


def test_bins_tail_fragment_order_replayed_with_the_mfma_lane_mapping():
    g = torch.Generator().manual_seed(2)
    for ctot in (168, 160):
        w0 = torch.randn(80, ctot, 1, 1, generator=g)
        mlp0 = pk.pack_conv(w0, torch.randn(80, generator=g), dtype=torch.float32)
        mlp2 = pk.pack_conv(torch.randn(4, 80, 1, 1, generator=g), torch.randn(4, generator=g), dtype=torch.float32)
        tw = pk.bins_tail_weights(mlp0, mlp2, 128)
        assert tw is not None and tw.nq == (ctot + 15) // 16 and tuple(tw.w0f.shape) == (tw.nq, 5, 64, 4)
        X = torch.randn(16, tw.nq * 16, generator=g).numpy().astype(np.float64)      # 16 pixels x padded channels
        X[:, ctot:] = 0.0
        lane = np.arange(64)
        r, gq = lane & 15, lane >> 4
        w0f = tw.w0f.numpy().astype(np.float64)
        want = X[:, :ctot] @ w0[:, :, 0, 0].numpy().astype(np.float64).T                # [px, 80]
        for f in range(5):
            acc = np.zeros((64, 4))
            for q in range(tw.nq):
                for e in range(4):
                    a = w0f[q, f, :, e]                                              # lane (i = r, k = g): W0[16 f + i][16 q + 4 k + e]
                    b = X[r, 16 * q + 4 * gq + e]                                    # lane (j = r, k = g): X[px j][16 q + 4 k + e]
                    acc = mfma_16x16x4(a, b, acc)
            # D[i][j] in lane j + 16 (i // 4), register i % 4  ->  lane (r = px, g): channels 16 f + 4 g + e'
            for l in range(64):
                for e in range(4):
                    assert abs(acc[l, e] - want[l & 15, 16 * f + 4 * (l >> 4) + e]) < 1e-9
        assert bool((tw.w2 == mlp2.w[:4, :80]).all()) and bool((tw.b0 == mlp0.bias[:80]).all())
    # shapes the kernel is not written for are refused (the engine then keeps the four launches)
    assert pk.bins_tail_weights(mlp0, mlp2, 64) is None
    assert pk.bins_tail_weights(pk.pack_conv(torch.randn(80, 168, 1, 1), torch.zeros(80), dtype=torch.bfloat16), mlp2, 128) is None
