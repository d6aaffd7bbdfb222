"""Device-side output stage: what ``Tester.run`` (estimator/tester/tester.py:66-84) does on the
CPU with the stitched depth map - colour rendering, 16-bit export, evaluation metrics - as HIP kernels on the
depth tensor where it already lives.  Same names and arguments as the reference functions; results come back as
small device tensors / python floats (one 104-byte D2H for the metrics).  There is no CPU path.
"""
import math

import numpy as np
import torch

_LUTS = {}


def _ops():
    from .hip_ops import ops        # fails loudly when the HIP extension is missing
    return ops


def _plane_any(value):
    """squeeze to one [H,W] plane, dtype kept (masks)"""
    v = value.detach()
    while v.dim() > 2 and v.shape[0] == 1:
        v = v[0]
    if v.dim() != 2:
        raise ValueError(f"expected one map, got shape {tuple(value.shape)}")
    return v


def colormap_lut(cmap, device):
    """(N+3, 4) uint8 table of a matplotlib colormap (N colours + under / over / bad rows), scaled like
    Colormap.__call__(bytes=True); built once per (cmap, device) on the host - it is 1 KiB of constants."""
    key = (cmap, str(device))
    if key not in _LUTS:
        import matplotlib
        cm = matplotlib.colormaps[cmap] if hasattr(matplotlib, "colormaps") else matplotlib.cm.get_cmap(cmap)
        if not cm._isinit:
            cm._init()
        _LUTS[key] = (torch.from_numpy((cm._lut * 255).astype(np.uint8)).to(device).contiguous(), int(cm.N))
    return _LUTS[key]


def _plane(value):
    v = value.detach()
    while v.dim() > 2 and v.shape[0] == 1:
        v = v[0]
    if v.dim() != 2:
        raise ValueError(f"expected one depth map, got shape {tuple(value.shape)}")
    return v.float().contiguous()


def gamma_table():
    """color.py:86-91 on one byte: (img / 255) ** 2.2 * 255 in float64, truncated by astype(uint8)."""
    return (np.power(np.arange(256) / 255, 2.2) * 255).astype(np.uint8)


def colorize(value, vmin=None, vmax=None, cmap="turbo_r", invalid_val=-99, invalid_mask=None, background_color=(128, 128, 128, 255),
             gamma_corrected=False, value_transform=None, vminp=2, vmaxp=95, ops=None):
    """estimator/utils/color.py:95-150 on the device: -> uint8 tensor [H,W,4] (RGBA).  vmin / vmax default to the
    exact 2nd / 95th percentile of the valid pixels (radix select, no sort, no host round trip).

    invalid_mask (bool, same grid; numpy or tensor) replaces the `value == invalid_val` test as in the reference (:121-122).
    gamma_corrected is a per-byte function (:86-91), so it is applied to the 1 KiB colour table and the background colour, not to the
    image.  value_transform is an arbitrary python callable on the normalised numpy array (:140-141): only in that case the
    normalised plane makes one host round trip (the reference hands the callable a numpy array with NaN at the invalid pixels)."""
    ops = ops or _ops()
    d = _plane(value)
    m = None
    if invalid_mask is not None:
        m = torch.as_tensor(np.asarray(invalid_mask.detach().cpu()) if isinstance(invalid_mask, torch.Tensor) else np.asarray(invalid_mask))
        m = (m.reshape(d.shape) != 0).to(torch.uint8).to(d.device).contiguous()
    if vmin is None or vmax is None:
        vmm = ops.percentiles(d, vminp, vmaxp, invalid_val=invalid_val, invalid_mask=m)
        if vmin is not None:
            vmm[0] = float(vmin)
        if vmax is not None:
            vmm[1] = float(vmax)
    else:
        vmm = torch.tensor([float(vmin), float(vmax)], dtype=torch.float32).to(d.device)
    lut, N = colormap_lut(cmap, d.device)
    bg = tuple(int(v) & 255 for v in background_color)
    if gamma_corrected:
        g = gamma_table()
        lut = torch.from_numpy(g[lut.cpu().numpy()]).to(d.device).contiguous()
        bg = tuple(int(g[v]) for v in bg)
    out = torch.empty(d.shape + (4,), dtype=torch.uint8, device=d.device)
    if value_transform is not None:
        if m is None:
            m = (d == invalid_val).to(torch.uint8) if invalid_val is not None else torch.zeros_like(d, dtype=torch.uint8)
        lo, hi = (np.float32(t) for t in vmm.tolist())
        x = d.cpu().numpy()
        x = (x - lo) / (hi - lo) if lo != hi else x * np.float32(0)
        x[m.cpu().numpy() != 0] = np.nan
        x = np.ascontiguousarray(np.asarray(value_transform(x), dtype=np.float32))
        d = torch.from_numpy(x).to(d.device)
        vmm = torch.tensor([0.0, 1.0], dtype=torch.float32).to(d.device)        # (x - 0) / (1 - 0) == x exactly
    return ops.colorize(d, vmm, lut, N, invalid_val, bg, out, invalid_mask=m)


def depth_to_uint16(depth, ops=None):
    """tester.py:75: (depth * 256).astype('uint16') -> torch.uint16 [H,W] on the device."""
    ops = ops or _ops()
    d = _plane(depth)
    return ops.depth_to_u16(d, torch.empty(d.shape, dtype=torch.uint16, device=d.device))


def crop_rectangle(gh, gw, garg_crop, eigen_crop, dataset):
    """estimator/utils/metric.py:113-126 -> rows [y0,y1), cols [x0,x1) of the evaluation mask."""
    if garg_crop:
        return int(0.40810811 * gh), int(0.99189189 * gh), int(0.03594771 * gw), int(0.96405229 * gw)
    if eigen_crop:
        if dataset == "kitti":
            return int(0.3324324 * gh), int(0.91351351 * gh), int(0.0359477 * gw), int(0.96405229 * gw)
        return 45, 471, 41, 601
    return 0, gh, 0, gw


def metrics_from_sums(s):
    """13 accumulated sums (include/pf_hip.h: pf_depth_metrics) -> the reference's metric dict (metric.py:30-52,136-146)."""
    n = s[0]
    if n <= 0:
        nan = float("nan")
        r = dict(a1=nan, a2=nan, a3=nan, abs_rel=nan, rmse=nan, log_10=nan, rmse_log=nan, silog=nan, sq_rel=nan)
    else:
        var = s[9] / n - (s[8] / n) ** 2
        r = dict(a1=s[1] / n, a2=s[2] / n, a3=s[3] / n, abs_rel=s[4] / n, rmse=math.sqrt(s[6] / n), log_10=s[10] / n,
                 rmse_log=math.sqrt(s[7] / n), silog=(math.sqrt(var) if var >= 0 else float("nan")) * 100, sq_rel=s[5] / n)
    return r


def compute_metrics(gt, pred, interpolate=True, garg_crop=False, eigen_crop=True, dataset="nyu", min_depth_eval=0.1, max_depth_eval=10,
                    disp_gt_edges=None, additional_mask=None, ops=None):
    """estimator/utils/metric.py:87-148 on the device (signature and defaults of the reference).  pred is resized to
    the ground-truth grid inside the kernel when `interpolate` and the grids differ."""
    ops = ops or _ops()
    g, p = _plane(gt), _plane(pred)
    if g.shape != p.shape and not interpolate:
        raise ValueError("gt and pred grids differ and interpolate=False")
    e = None
    if disp_gt_edges is not None:
        e = _plane(disp_gt_edges.to(g.device))
    am = None
    if additional_mask is not None:                                           # metric.py:128-130 (prompt-depth evaluation)
        am = (_plane_any(additional_mask.to(g.device)) != 0).to(torch.uint8).contiguous()
    out = torch.empty(13, dtype=torch.float64, device=g.device)
    ops.depth_metrics(g, p, e, min_depth_eval, max_depth_eval, crop_rectangle(g.shape[0], g.shape[1], garg_crop, eigen_crop, dataset), out,
                      additional_mask=am)
    s = out.cpu().tolist()
    r = metrics_from_sums(s)
    if disp_gt_edges is not None:
        r["see"] = s[11] / s[12] if s[12] > 0 else 0.0
    return r
