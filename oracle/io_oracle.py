"""CPU restatement of the reference's input / output side of the tiled-inference path.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).  Each function cites the reference
file:line it follows.  Pinning: `oracle/make_golden_io.py` imports the reference's own `colorize`,
`compute_metrics` and `read_image` arithmetic (through `oracle/ref_shim.py`, in the build container) and commits
their outputs as `tests/golden/io_side.npz`; `tests/test_oracle_io.py` checks this restatement against those
fixtures.  Two third-party behaviours are restated rather than imported:
  * `np.percentile` - the reference pins numpy 1.24.4 (environment.yml:13), this image has numpy 2.2: the two differ
    in the dtype of the virtual index (float64 vs float32 for float32 input).  `percentile_linear` follows 1.24.4
    (index and weight in float64, difference in float32) - PARITY UNPINNED against 1.24.4 itself; the KAT in
    tests/test_oracle_io.py bounds its distance to the installed numpy by 4 float32 ulps.
  * matplotlib `Colormap.__call__(X, bytes=True)` - restated in `colormap_bytes` (matplotlib 3.7.3 pinned,
    3.10 here, same rules) and checked against the installed matplotlib on random input.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- input side
def read_image_arith(img_u8_rgb, image_resolution, dataset_name="general"):
    """general_dataset.py:22-47 after the decode: uint8 HWC (RGB, or the raw BGR file order for 'u4k') -> HWC float.
    'u4k' (:24-25): /255.0, float32, reverse channels.  Otherwise (:27-33, :40-46): /255.0 stays float64 through
    F.interpolate(bicubic, align_corners=True)."""
    if dataset_name == "u4k":
        img = img_u8_rgb / 255.0
        return img.astype(np.float32)[:, :, ::-1].copy()
    img = img_u8_rgb / 255.0
    t = F.interpolate(torch.tensor(img).unsqueeze(dim=0).permute(0, 3, 1, 2), tuple(image_resolution), mode="bicubic",
                      align_corners=True)
    return t.squeeze().permute(1, 2, 0).numpy()


def dataset_item(img_hwc, process_shape):
    """general_dataset.py:201-202: image_hr = to_tensor(img).float() (HWC -> CHW, transformers/format.py:5-14);
    image_lr = Resize(image_hr[None]) = bilinear align_corners=True to the network input (depth_anything/transform.py:127-129)."""
    image_hr = torch.from_numpy(np.ascontiguousarray(img_hwc.transpose((2, 0, 1)))).float()
    image_lr = F.interpolate(image_hr.unsqueeze(0), tuple(process_shape), mode="bilinear", align_corners=True).squeeze(0)
    return image_hr, image_lr


# ---------------------------------------------------------------- output side
def percentile_linear(values, q):
    """np.percentile(values, q) for 1-D float32 `values`, numpy 1.24.4 semantics (lib/function_base.py _quantile
    'linear' + _lerp): virtual index (n-1)*q/100 and gamma in float64; previous/next order statistics float32;
    diff = next - previous in float32; previous + diff*gamma in float64, next - diff*(1-gamma) where gamma >= 0.5.
    Returns float32 (the value the float32 image arithmetic of colorize() sees)."""
    v = np.sort(np.asarray(values, dtype=np.float32).ravel())
    n = v.shape[0]
    if n == 0:
        return np.float32(np.nan)
    vi = (n - 1) * (np.float64(q) / 100.0)
    lo = int(np.floor(vi))
    hi = min(lo + 1, n - 1)
    g = np.float64(vi - lo)
    a, b = v[lo], v[hi]
    diff = np.float32(b - a)
    r = np.float64(a) + np.float64(diff) * g
    if g >= 0.5:
        r = np.float64(b) - np.float64(diff) * (1.0 - g)
    return np.float32(r)


def colormap_lut_bytes(cmap_name):
    """(N+3, 4) uint8 table of a matplotlib colormap: N colours, then the under / over / bad rows, scaled exactly as
    Colormap.__call__(bytes=True) does ((lut * 255).astype(np.uint8))."""
    import matplotlib
    cm = matplotlib.colormaps[cmap_name] if hasattr(matplotlib, "colormaps") else matplotlib.cm.get_cmap(cmap_name)
    if not cm._isinit:
        cm._init()
    return (cm._lut * 255).astype(np.uint8), cm.N


def colormap_bytes(x, lut, N):
    """matplotlib Colormap.__call__(x, bytes=True) for a float32 array (colors.py): xa = x*N; xa<0 -> -1; xa==N -> N-1;
    clip to [-1, N]; astype(int); > N-1 -> over (N+1); < 0 -> under (N); nan -> bad (N+2)."""
    x = np.asarray(x, dtype=np.float32)
    bad = np.isnan(x)
    with np.errstate(invalid="ignore"):
        xa = x * np.float32(N)
        xa[xa < 0] = -1
        xa[xa == N] = N - 1
        np.clip(xa, -1, N, out=xa)
        xa = np.where(bad, 0, xa).astype(int)
    xa[xa > N - 1] = N + 1
    xa[xa < 0] = N
    xa[bad] = N + 2
    return lut[xa]


def colorize(value, vmin=None, vmax=None, cmap="turbo_r", invalid_val=-99, invalid_mask=None, background_color=(128, 128, 128, 255),
             gamma_corrected=False, value_transform=None, vminp=2, vmaxp=95):
    """estimator/utils/color.py:95-150 -> (H, W, 4) uint8.
    tester.py:68-71 calls it with cmap 'magma_r' / 'gray_r' and takes [:, :, [2, 1, 0]]."""
    value = np.asarray(value, dtype=np.float32).squeeze().copy()
    if invalid_mask is None:                                        # :121-122
        invalid_mask = value == invalid_val
    invalid_mask = np.asarray(invalid_mask, dtype=bool)
    mask = np.logical_not(invalid_mask)
    vmin = percentile_linear(value[mask], vminp) if vmin is None else np.float32(vmin)
    vmax = percentile_linear(value[mask], vmaxp) if vmax is None else np.float32(vmax)
    if vmin != vmax:
        value = (value - vmin) / (vmax - vmin)
    else:
        value = value * np.float32(0.)
    value[invalid_mask] = np.nan
    if value_transform:                                             # :140-141
        value = value_transform(value)
    lut, N = colormap_lut_bytes(cmap)
    img = colormap_bytes(value, lut, N)
    img[invalid_mask] = background_color
    if gamma_corrected:                                             # :86-91 (all four channels, alpha included)
        img = img / 255
        img = np.power(img, 2.2)
        img = img * 255
        img = img.astype(np.uint8)
    return img


def depth_to_uint16(depth):
    """tester.py:75: (result.squeeze().cpu().numpy() * 256).astype('uint16')"""
    return (np.asarray(depth, dtype=np.float32).squeeze() * 256).astype("uint16")


def compute_errors(gt, pred):
    """estimator/utils/metric.py:10-52 verbatim arithmetic (float32 in, numpy means)."""
    thresh = np.maximum((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).mean()
    a2 = (thresh < 1.25 ** 2).mean()
    a3 = (thresh < 1.25 ** 3).mean()
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    err = np.log(pred) - np.log(gt)
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    log_10 = (np.abs(np.log10(gt) - np.log10(pred))).mean()
    return dict(a1=a1, a2=a2, a3=a3, abs_rel=abs_rel, rmse=rmse, log_10=log_10, rmse_log=rmse_log, silog=silog, sq_rel=sq_rel)


def soft_edge_error(pred, gt, radius=1):
    """metric.py:54-71: min over the (2r+1)^2 zero-filled shifts of gt of |shift(gt) - pred|."""
    H, W = gt.shape
    pad = np.zeros((H + 2 * radius, W + 2 * radius), dtype=gt.dtype)
    pad[radius:radius + H, radius:radius + W] = gt
    out = None
    for dx in range(-radius, radius + 1):
        for dy in range(-radius, radius + 1):
            sh = pad[radius - dy:radius - dy + H, radius - dx:radius - dx + W]      # shifted[y][x] = gt[y-dy][x-dx]
            d = np.abs(sh - pred)
            out = d if out is None else np.minimum(out, d)
    return out


def compute_metrics(gt, pred, min_depth_eval=0.1, max_depth_eval=10, disp_gt_edges=None, garg_crop=False, eigen_crop=False,
                    dataset="nyu", additional_mask=None):
    """metric.py:87-148 (interpolate=True).  gt, pred: torch tensors [.., H, W] / [1, 1, h, w]."""
    if gt.shape[-2:] != pred.shape[-2:]:
        pred = F.interpolate(pred, gt.shape[-2:], mode="bilinear", align_corners=False).squeeze()
    pred = pred.squeeze().cpu().numpy().copy()
    pred[pred < min_depth_eval] = min_depth_eval
    pred[pred > max_depth_eval] = max_depth_eval
    pred[np.isinf(pred)] = max_depth_eval
    pred[np.isnan(pred)] = min_depth_eval
    gt_depth = gt.squeeze().cpu().numpy()
    valid_mask = np.logical_and(gt_depth > min_depth_eval, gt_depth < max_depth_eval)
    eval_mask = np.ones(valid_mask.shape)
    if garg_crop or eigen_crop:
        gh, gw = gt_depth.shape
        eval_mask = np.zeros(valid_mask.shape)
        y0, y1, x0, x1 = crop_rectangle(gh, gw, garg_crop, eigen_crop, dataset)
        eval_mask[y0:y1, x0:x1] = 1
    valid_mask = np.logical_and(valid_mask, eval_mask)
    if additional_mask is not None:                                 # :128-130
        valid_mask = np.logical_and(valid_mask, additional_mask.squeeze().detach().cpu().numpy())
    metrics = compute_errors(gt_depth[valid_mask], pred[valid_mask])
    if disp_gt_edges is not None:
        edges = disp_gt_edges.squeeze().numpy()
        mask = np.logical_and(valid_mask.squeeze(), edges)
        see = 0.0
        if mask.sum() > 0:
            see = soft_edge_error(pred, gt_depth)[mask].mean()
        metrics["see"] = see
    return metrics


def crop_rectangle(gh, gw, garg_crop, eigen_crop, dataset):
    """metric.py:113-126: rows [y0,y1) x cols [x0,x1) of the evaluation mask."""
    if garg_crop:
        return int(0.40810811 * gh), int(0.99189189 * gh), int(0.03594771 * gw), int(0.96405229 * gw)
    if eigen_crop:
        if dataset == "kitti":
            return int(0.3324324 * gh), int(0.91351351 * gh), int(0.0359477 * gw), int(0.96405229 * gw)
        return 45, 471, 41, 601
    return 0, gh, 0, gw
