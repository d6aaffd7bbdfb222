"""Lane-level numpy model of csrc/imageops.hip bins_tail_kernel (test infrastructure): the per-K-group operand build with its one-group
look-ahead (which source feeds which group), the MFMA lane mapping against packing.bins_tail_weights, the split of the 80 hidden channels
and of the 64 bins over the four lane groups with their xor-shuffle reductions -- replayed for a few pixels and compared with the four
separate steps the kernel replaces (tests/fake_ops.py: resize, two 1x1 convolutions, log-binomial expectation)."""
import math

import numpy as np
import pytest
import torch

from patchfusion_amd import packing as pk
from tests.fake_ops import ops as ref_ops
from tests.wino_fused_model import mfma_16x16x4


def _lerp(dst, scale, n):
    src = np.float32(scale) * np.float32(dst)
    i0 = int(src)
    i1 = i0 + (1 if i0 < n - 1 else 0)
    l1 = float(np.float32(src) - np.float32(i0))
    return i0, i1, 1.0 - l1, l1


def _scale(n_in, n_out):
    return np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)


@pytest.mark.parametrize("ctot", [168, 160])
def test_bins_tail_kernel_model_equals_the_four_steps(ctot):
    g = torch.Generator().manual_seed(5)
    B, H, W, he, we = 1, 5, 7, 3, 4
    mlp0 = pk.pack_conv(torch.randn(80, ctot, 1, 1, generator=g) / ctot ** 0.5, torch.randn(80, generator=g) * 0.1, dtype=torch.float32)
    mlp2 = pk.pack_conv(torch.randn(4, 80, 1, 1, generator=g) / 80 ** 0.5, torch.randn(4, generator=g) * 0.1, dtype=torch.float32)
    tw = pk.bins_tail_weights(mlp0, mlp2, 128)
    clb = torch.randn(B, H, W, ctot, generator=g)
    emb = torch.randn(B, he, we, 128, generator=g)
    cen = (torch.rand(B, he, we, 64, generator=g) * 5 + 0.5).sort(-1).values.contiguous()
    want = torch.zeros(B, H, W)
    ref_ops.bins_tail(clb, emb, tw, cen, want, 0.0212, 50.0)

    nq, w0f, b0, w2, b2 = tw.nq, tw.w0f.numpy().astype(np.float64), tw.b0.numpy().astype(np.float64), tw.w2.numpy().astype(np.float64), tw.b2.numpy().astype(np.float64)
    clbn, embn, cenn = clb.numpy().astype(np.float64), emb.numpy().astype(np.float64), cen.numpy().astype(np.float64)
    lane = np.arange(64)
    r, gq = lane & 15, lane >> 4
    total = B * H * W
    got = np.zeros(total)
    she, swe = _scale(he, H), _scale(we, W)
    n_ = 63.0 + 1e-7
    logc = [n_ * math.log(n_) - (k + 1e-7) * math.log(k + 1e-7) - (n_ - (k + 1e-7)) * math.log(n_ - (k + 1e-7) + 1e-7) for k in range(64)]
    for grp in range((total + 15) // 16):
        pix = np.minimum(grp * 16 + r, total - 1)
        ox, oy, b = pix % W, (pix // W) % H, pix // (W * H)
        # ---- operand per K group, with the kernel's look-ahead order: q = 0, 1 last; 2..9 embedding group q - 2; 10 rel (lanes g < 2)
        X = np.zeros((64, 11, 4))
        for l in range(64):
            for q in range(nq):
                c0 = 4 * gq[l]
                if q < 2:
                    X[l, q] = clbn[b[l], oy[l], ox[l], 16 * q + c0:16 * q + c0 + 4]
                elif q < 10:
                    y0, y1, ly0, ly1 = _lerp(oy[l], she, he)
                    x0, x1, lx0, lx1 = _lerp(ox[l], swe, we)
                    ch = slice(16 * (q - 2) + c0, 16 * (q - 2) + c0 + 4)
                    X[l, q] = ly0 * (lx0 * embn[b[l], y0, x0, ch] + lx1 * embn[b[l], y0, x1, ch]) + ly1 * (lx0 * embn[b[l], y1, x0, ch] + lx1 * embn[b[l], y1, x1, ch])
                elif gq[l] < 2:
                    X[l, q] = clbn[b[l], oy[l], ox[l], tw.rel_off + c0:tw.rel_off + c0 + 4]
        # ---- 80 x K on the MFMA: step (q, e): A = packed weights component e, B = X[., q, e]
        acc = np.zeros((5, 64, 4))
        for q in range(nq):
            for e in range(4):
                for f in range(5):
                    acc[f] = mfma_16x16x4(w0f[q, f, :, e], X[:, q, e], acc[f])
        # ---- bias + GELU; 80 -> 4 as per-lane partial sums + xor-shuffles over the lane groups; Softplus
        o4 = np.zeros((64, 4))
        for l in range(64):
            for f in range(5):
                for e in range(4):
                    ch = 16 * f + 4 * gq[l] + e
                    v = acc[f, l, e] + b0[ch]
                    t = 0.5 * v * (1.0 + math.erf(v * 0.7071067811865476))
                    o4[l] += w2[:, ch] * t
        o4 = o4 + o4[lane ^ 16]
        o4 = o4 + o4[lane ^ 32]
        o4 = o4 + b2
        o4 = np.where(o4 > 20.0, o4, np.log1p(np.exp(np.minimum(o4, 20.0))))
        # ---- log-binomial: 16 bins per lane group, max / sums reduced across the groups
        p0, p1, t0, t1 = (o4[:, i] + 1e-4 for i in range(4))
        p = p0 / (p0 + p1)
        t = (50.0 - 0.0212) * (t0 / (t0 + t1)) + 0.0212
        omp = np.clip(1.0 - p, 1e-4, 1.0)
        p = np.clip(p, 1e-4, 1.0)
        lp, lq = np.log(p), np.log(omp)
        yk = np.zeros((64, 16))
        for l in range(64):
            for j in range(16):
                k = 16 * gq[l] + j
                yk[l, j] = (logc[k] + k * lp[l] + (63 - k) * lq[l]) / t[l]
        mx = yk.max(1)
        mx = np.maximum(mx, mx[lane ^ 16])
        mx = np.maximum(mx, mx[lane ^ 32])
        shc, swc = _scale(he, H), _scale(we, W)
        num, den = np.zeros(64), np.zeros(64)
        for l in range(64):
            y0, y1, ly0, ly1 = _lerp(oy[l], shc, he)
            x0, x1, lx0, lx1 = _lerp(ox[l], swc, we)
            for j in range(16):
                k = 16 * gq[l] + j
                pe = math.exp(yk[l, j] - mx[l])
                c = ly0 * (lx0 * cenn[b[l], y0, x0, k] + lx1 * cenn[b[l], y0, x1, k]) + ly1 * (lx0 * cenn[b[l], y1, x0, k] + lx1 * cenn[b[l], y1, x1, k])
                num[l] += pe * c
                den[l] += pe
        num = num + num[lane ^ 16]; num = num + num[lane ^ 32]
        den = den + den[lane ^ 16]; den = den + den[lane ^ 32]
        for l in range(16):                      # lanes g == 0 store
            if grp * 16 + l < total:
                got[grp * 16 + l] = num[l] / den[l]
    err = np.abs(got.reshape(B, H, W) - want.numpy()).max()
    assert err < 5e-5, err
