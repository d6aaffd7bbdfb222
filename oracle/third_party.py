"""TEST INFRASTRUCTURE ONLY (oracle). Never imported by the product path (patchfusion_amd/).

CPU restatements of the two third-party leaves whose arithmetic is NOT in /root/reference:

* ``roi_align``      -- torchvision 0.16.2 (reference environment.yml:21), call sites
                        estimator/models/patchfusion.py:232,235,247,251 and
                        estimator/models/blocks/guided_fusion_model.py:202, always
                        ``aligned=True`` and the default ``sampling_ratio=-1``.
                        Restated from the published semantics of torchvision's
                        ``roi_align_forward_kernel_impl`` / ``bilinear_interpolate``
                        (torchvision/csrc/ops/cpu/roi_align_kernel.cpp).
* ``gaussian_blur``  -- opencv-python 4.8.1.78 ``cv2.GaussianBlur`` (environment.yml:30), call
                        site estimator/models/utils.py:44: separable kernel
                        exp(-x^2/(2 sigma^2)) normalised to sum 1, BORDER_REFLECT_101.

PARITY UNPINNED for these two leaves: the reference holds no test/golden vector for them and
neither torchvision nor OpenCV is installed in this image.  They are pinned instead by analytic
known-answer tests (tests/test_oracle_third_party.py): identity ROI, half-pixel shift, constant
maps, out-of-range zeroing; mask symmetry / normalisation / interior==1.
"""
import math

import numpy as np
import torch


# --------------------------------------------------------------------------------------------
# torchvision.ops.roi_align
# --------------------------------------------------------------------------------------------
def _f32(x):
    return np.float32(x)


def roi_align_sample_grid(rois, out_h, out_w, spatial_scale, in_h, in_w, aligned=True, sampling_ratio=-1):
    """Return per-roi sampling description in float32 arithmetic (the same op order as the
    torchvision kernel, which computes in T=float): (start_h, start_w, bin_h, bin_w, grid_h, grid_w).
    """
    scale = _f32(spatial_scale)
    offset = _f32(0.5) if aligned else _f32(0.0)
    out = []
    for r in np.asarray(rois, dtype=np.float32):
        x1, y1, x2, y2 = r[-4:]
        start_w = _f32(x1 * scale) - offset
        start_h = _f32(y1 * scale) - offset
        end_w = _f32(x2 * scale) - offset
        end_h = _f32(y2 * scale) - offset
        roi_w = _f32(end_w - start_w)
        roi_h = _f32(end_h - start_h)
        if not aligned:
            roi_w = max(roi_w, _f32(1.0))
            roi_h = max(roi_h, _f32(1.0))
        bin_h = _f32(roi_h / _f32(out_h))
        bin_w = _f32(roi_w / _f32(out_w))
        grid_h = sampling_ratio if sampling_ratio > 0 else int(math.ceil(_f32(roi_h / _f32(out_h))))
        grid_w = sampling_ratio if sampling_ratio > 0 else int(math.ceil(_f32(roi_w / _f32(out_w))))
        out.append((start_h, start_w, bin_h, bin_w, grid_h, grid_w))
    return out


def roi_align(inp, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """inp [N,C,H,W] float32, rois [K,5] (batch_idx, x1, y1, x2, y2) -> [K,C,oh,ow].

    Vectorised over (C, oh, ow); loops over rois and the (small) sampling grid.
    """
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    out_h, out_w = int(output_size[0]), int(output_size[1])
    inp = inp.float()
    N, C, H, W = inp.shape
    rois_np = rois.detach().cpu().numpy().astype(np.float32)
    K = rois_np.shape[0]
    out = torch.zeros((K, C, out_h, out_w), dtype=torch.float32, device=inp.device)
    grids = roi_align_sample_grid(rois_np, out_h, out_w, spatial_scale, H, W, aligned, sampling_ratio)
    ph = torch.arange(out_h, dtype=torch.float32, device=inp.device)
    pw = torch.arange(out_w, dtype=torch.float32, device=inp.device)
    for k in range(K):
        b = int(rois_np[k, 0])
        start_h, start_w, bin_h, bin_w, gh, gw = grids[k]
        count = max(gh * gw, 1)
        acc = torch.zeros((C, out_h, out_w), dtype=torch.float32, device=inp.device)
        for iy in range(gh):
            # y = roi_start_h + ph * bin_size_h + (iy + .5) * bin_size_h / grid_h   (float32)
            y = float(start_h) + ph * float(bin_h) + float(_f32(_f32(iy + 0.5) * bin_h / _f32(gh)))
            for ix in range(gw):
                x = float(start_w) + pw * float(bin_w) + float(_f32(_f32(ix + 0.5) * bin_w / _f32(gw)))
                acc += _bilinear_gather(inp[b], y, x, H, W)
        out[k] = acc / float(count)
    return out


def _bilinear_gather(feat, y, x, H, W):
    """feat [C,H,W]; y [oh], x [ow] float32 sample coordinates -> [C,oh,ow].
    torchvision ``bilinear_interpolate``: zero outside [-1, size]; clamp at 0; clamp top edge."""
    vy = (y >= -1.0) & (y <= H)
    vx = (x >= -1.0) & (x <= W)
    y = y.clamp(min=0.0)
    x = x.clamp(min=0.0)
    y_low = y.floor().long()
    x_low = x.floor().long()
    top_y = y_low >= H - 1
    top_x = x_low >= W - 1
    y_low = torch.where(top_y, torch.full_like(y_low, H - 1), y_low)
    x_low = torch.where(top_x, torch.full_like(x_low, W - 1), x_low)
    y_high = torch.where(top_y, y_low, y_low + 1)
    x_high = torch.where(top_x, x_low, x_low + 1)
    y = torch.where(top_y, y_low.float(), y)
    x = torch.where(top_x, x_low.float(), x)
    ly = y - y_low.float()
    lx = x - x_low.float()
    hy = 1.0 - ly
    hx = 1.0 - lx
    v1 = feat[:, y_low][:, :, x_low]
    v2 = feat[:, y_low][:, :, x_high]
    v3 = feat[:, y_high][:, :, x_low]
    v4 = feat[:, y_high][:, :, x_high]
    w1 = hy[:, None] * hx[None, :]
    w2 = hy[:, None] * lx[None, :]
    w3 = ly[:, None] * hx[None, :]
    w4 = ly[:, None] * lx[None, :]
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    valid = (vy[:, None] & vx[None, :]).to(val.dtype)
    return val * valid


# --------------------------------------------------------------------------------------------
# cv2.GaussianBlur
# --------------------------------------------------------------------------------------------
def gaussian_kernel1d(ksize, sigma):
    """cv::getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0: exp(-(i-(k-1)/2)^2 / (2 sigma^2)),
    normalised to sum 1 (computed in double, stored as float32)."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * float(sigma) * float(sigma)))
    k /= k.sum()
    return k.astype(np.float32)


def gaussian_blur(img, ksize, sigma):
    """cv2.GaussianBlur(img, (k, k), sigma) for a 2-D float32 array, default border
    BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba).  Row pass then column pass."""
    kx, ky = int(ksize[0]), int(ksize[1])
    img = np.asarray(img, dtype=np.float32)
    kern_x = gaussian_kernel1d(kx, sigma).astype(np.float64)
    kern_y = gaussian_kernel1d(ky, sigma).astype(np.float64)
    rx, ry = kx // 2, ky // 2
    pad = np.pad(img.astype(np.float64), ((0, 0), (rx, rx)), mode="reflect")
    tmp = np.zeros(img.shape, dtype=np.float64)
    for i in range(kx):
        tmp += kern_x[i] * pad[:, i:i + img.shape[1]]
    tmp = tmp.astype(np.float32).astype(np.float64)  # OpenCV keeps a float32 intermediate buffer
    pad = np.pad(tmp, ((ry, ry), (0, 0)), mode="reflect")
    out = np.zeros(img.shape, dtype=np.float64)
    for i in range(ky):
        out += kern_y[i] * pad[i:i + img.shape[0], :]
    return out.astype(np.float32)


class Normalize:
    """torchvision.transforms.Normalize restated: (x - mean[c]) / std[c] on [...,C,H,W]."""

    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, x):
        return (x - self.mean.to(x.device)) / self.std.to(x.device)
