"""Analytic known-answer tests pinning the two 'parity unpinned' third-party restatements
(oracle/third_party.py): torchvision roi_align and cv2.GaussianBlur."""
import numpy as np
import torch

from oracle import third_party as tp
from oracle.pf_oracle import generatemask


def test_roi_align_identity_roi():
    # aligned=True, ROI == whole map, output == map size -> 1 sample per bin at pixel centres -> identity
    x = torch.randn(1, 3, 7, 9)
    rois = torch.tensor([[0, 0.0, 0.0, 9.0, 7.0]])
    y = tp.roi_align(x, rois, (7, 9), 1.0, aligned=True)
    assert torch.allclose(y, x, atol=1e-6)


def test_roi_align_half_pixel_shift_and_constant():
    x = torch.arange(20, dtype=torch.float32).view(1, 1, 4, 5)
    # shift the ROI by +0.5 px in x: every sample lies midway between two columns
    rois = torch.tensor([[0, 0.5, 0.0, 4.5, 4.0]])
    y = tp.roi_align(x, rois, (4, 4), 1.0, aligned=True)
    expect = 0.5 * (x[..., :4] + x[..., 1:5])
    assert torch.allclose(y, expect, atol=1e-6)
    c = torch.full((1, 2, 6, 6), 3.25)
    y = tp.roi_align(c, torch.tensor([[0, 1.0, 1.0, 4.0, 5.0]]), (6, 6), 1.0, aligned=True)
    assert torch.allclose(y, torch.full_like(y, 3.25), atol=1e-6)


def test_roi_align_quarter_roi_upsamples_one_sample_per_bin():
    # the PatchFusion use: ROI = 1/2 x 1/2 of the map resampled to the full grid (sampling grid = 1)
    x = torch.randn(1, 2, 8, 8)
    rois = torch.tensor([[0, 4.0, 4.0, 8.0, 8.0]])
    y = tp.roi_align(x, rois, (8, 8), 1.0, aligned=True)
    # sample coordinates: 4 - .5 + (i + .5) * .5 = 3.75 + i/2
    ys = 3.75 + 0.5 * torch.arange(8)
    g = tp._bilinear_gather(x[0], ys, ys, 8, 8)
    assert torch.allclose(y[0], g, atol=1e-6)
    # the last sample (7.25) clamps to the edge pixel (torchvision: y_low >= H-1 -> y = H-1)
    assert torch.allclose(y[0, :, -1, -1], x[0, :, -1, -1], atol=1e-6)


def test_roi_align_out_of_range_is_zero_and_scale_and_batch_index():
    x = torch.ones(2, 1, 4, 4)
    x[1] *= 2
    rois = torch.tensor([[1, -8.0, -8.0, -4.0, -4.0], [1, 0.0, 0.0, 8.0, 8.0]])
    y = tp.roi_align(x, rois, (2, 2), 0.5, aligned=True)
    assert torch.all(y[0] == 0)
    assert torch.allclose(y[1], torch.full_like(y[1], 2.0))


def test_roi_align_multi_sample_average():
    # ROI 4x4 pooled to 2x2 -> sampling grid 2x2 per bin -> mean of 4 bilinear samples
    x = torch.arange(16, dtype=torch.float32).view(1, 1, 4, 4)
    y = tp.roi_align(x, torch.tensor([[0, 0.0, 0.0, 4.0, 4.0]]), (2, 2), 1.0, aligned=True)
    expect = torch.tensor([[2.5, 4.5], [10.5, 12.5]])
    assert torch.allclose(y[0, 0], expect, atol=1e-6)


def test_gaussian_blur_properties():
    k = tp.gaussian_kernel1d(97, 24)
    assert abs(float(k.sum()) - 1) < 1e-6 and np.allclose(k, k[::-1]) and k.argmax() == 48
    assert np.isclose(k[48] / k[47], np.exp(1 / (2 * 24 * 24)), rtol=1e-6)
    c = np.full((20, 30), 2.5, np.float32)
    assert np.allclose(tp.gaussian_blur(c, (5, 5), 1.0), 2.5, atol=1e-6)
    # impulse response == outer product of the 1-D kernels (interior, away from the border)
    img = np.zeros((21, 21), np.float32)
    img[10, 10] = 1
    k5 = tp.gaussian_kernel1d(5, 1.0)
    out = tp.gaussian_blur(img, (5, 5), 1.0)
    assert np.allclose(out[8:13, 8:13], np.outer(k5, k5), atol=1e-7)
    # BORDER_REFLECT_101: the pixel next to the border is mirrored without repeating the border pixel
    row = np.zeros((1, 8), np.float32)
    row[0, 1] = 1
    k3 = tp.gaussian_kernel1d(3, 1.0)
    out = tp.gaussian_blur(row, (3, 1), 1.0)
    assert np.isclose(out[0, 0], 2 * k3[0], atol=1e-7)


def test_generatemask_properties():
    m = generatemask((112, 154))
    assert m.shape == (112, 154) and m.dtype == np.float32
    assert m.min() == 0.0 and m.max() == 1.0
    assert np.allclose(m, m[::-1, :], atol=1e-6) and np.allclose(m, m[:, ::-1], atol=1e-6)
    assert m[56, 77] == 1.0 and m[0, 0] < 1e-3
