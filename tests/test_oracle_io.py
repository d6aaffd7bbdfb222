"""Pins oracle/io_oracle.py (the CPU restatement of the reference's input / output side)
against tests/golden/io_side.npz, which oracle/make_golden_io.py produced by running the reference's own
read_image / colorize / compute_metrics in the build container."""
import os

import numpy as np
import torch

from oracle import io_oracle as io

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "io_side.npz"))


def test_read_image_arithmetic_matches_reference():
    a = io.read_image_arith(G["img_u8"], (96, 128))
    assert a.dtype == np.float64 and np.array_equal(a, G["read_image_96x128"])
    assert np.array_equal(io.read_image_arith(G["img_u8"], (61, 83)), G["read_image_same"])
    # identity size is the plain /255 conversion
    assert np.array_equal(G["read_image_same"], G["img_u8"] / 255.0)
    hr, lr = io.dataset_item(a, (28, 37))
    assert hr.shape == (3, 96, 128) and hr.dtype == torch.float32 and lr.shape == (3, 28, 37)


def _ulps(a, b):
    a, b = np.float32(a), np.float32(b)
    return abs(int(a.view(np.int32)) - int(b.view(np.int32)))


def test_percentile_restatement_vs_installed_numpy():
    d = G["depth"]
    v = d[d != -99]
    for q, ref in zip((2, 95), G["np_percentiles"]):
        assert _ulps(io.percentile_linear(v, q), ref) <= 4
    rs = np.random.RandomState(0)
    for n in (1, 2, 7, 1000, 65537):
        x = (rs.randn(n) * 3).astype(np.float32)
        for q in (0, 2, 50, 95, 100):
            assert _ulps(io.percentile_linear(x, q), np.percentile(x, q)) <= 4, (n, q)
    # exact cases: order statistics themselves
    x = np.arange(101, dtype=np.float32)
    assert io.percentile_linear(x, 2) == 2.0 and io.percentile_linear(x, 95) == 95.0


def test_colormap_restatement_vs_matplotlib():
    import matplotlib
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.rand(5000) * 1.4 - 0.2, [0.0, 1.0, -0.0, 1.0000001, np.nan, np.inf, -np.inf, 255 / 256, 0.99999994]]).astype(np.float32)
    for name in ("magma_r", "gray_r", "turbo_r"):
        lut, N = io.colormap_lut_bytes(name)
        assert lut.shape == (N + 3, 4) and N == 256
        with np.errstate(invalid="ignore"):
            ref = matplotlib.colormaps[name](x, bytes=True)
        assert np.array_equal(io.colormap_bytes(x, lut, N), ref), name


def test_colorize_matches_reference():
    d = G["depth"]
    for cmap in ("magma_r", "gray_r"):
        ref = G[f"colorize_{cmap}"]
        vmin, vmax = (np.float32(v) for v in G["np_percentiles"])
        img = io.colorize(d, vmin=vmin, vmax=vmax, cmap=cmap)          # the installed numpy's percentiles: exact
        assert img.dtype == np.uint8 and np.array_equal(img, ref)
        img = io.colorize(d, cmap=cmap)                                  # own (numpy 1.24 semantics) percentiles
        diff = np.abs(img.astype(int) - ref.astype(int)).max(axis=-1)
        assert (diff > 0).mean() < 2e-3
    assert (G["colorize_magma_r"][d == -99] == np.array([128, 128, 128, 255])).all()


def test_colorize_optional_arguments_match_reference():
    """color.py:121-122 invalid_mask, :140-141 value_transform, :86-91 gamma_corrected -- fixtures made by the reference's colorize"""
    d, im = G["depth"], G["invalid_mask"]
    lo, hi = (np.float32(v) for v in G["np_percentiles"])
    lom, him = (np.float32(v) for v in G["np_percentiles_mask"])
    assert np.array_equal(io.colorize(d, vmin=lom, vmax=him, cmap="magma_r", invalid_mask=im), G["colorize_mask"])
    assert np.array_equal(io.colorize(d, vmin=lo, vmax=hi, cmap="magma_r", gamma_corrected=True), G["colorize_gamma"])
    assert np.array_equal(io.colorize(d, vmin=lo, vmax=hi, cmap="gray_r", value_transform=np.square), G["colorize_transform"])
    assert np.array_equal(io.colorize(d, vmin=lom, vmax=him, cmap="turbo_r", invalid_mask=im, gamma_corrected=True, value_transform=np.square,
                                      background_color=(10, 200, 30, 255)), G["colorize_all"])
    # with an explicit mask the -99 pixels outside it are ordinary values (far below vmin -> the colormap's "under" colour)
    out = (d == -99) & ~im
    assert out.any() and (G["colorize_mask"][out] != np.array([128, 128, 128, 255])).any(axis=-1).all()
    # gamma touches the background and the alpha channel too: 128 -> int((128/255)**2.2*255) = 55, 255 -> 255
    assert (G["colorize_gamma"][d == -99] == np.array([55, 55, 55, 255])).all()
    # own percentiles (numpy 1.24 semantics) with the mask
    diff = np.abs(io.colorize(d, cmap="magma_r", invalid_mask=im).astype(int) - G["colorize_mask"].astype(int)).max(axis=-1)
    assert (diff > 0).mean() < 2e-3


def test_uint16_matches_reference():
    assert np.array_equal(io.depth_to_uint16(np.abs(G["depth"])), G["uint16"])


def test_compute_metrics_matches_reference():
    gt, pred, edges = (torch.from_numpy(G[k]) for k in ("gt", "pred", "edges"))
    r = io.compute_metrics(gt[None, None], pred[None, None], min_depth_eval=1e-3, max_depth_eval=80, disp_gt_edges=edges[None])
    keys = [str(k) for k in G["metrics_same_keys"]]
    assert sorted(r) == keys
    np.testing.assert_allclose([float(r[k]) for k in keys], G["metrics_same"], rtol=1e-6)
    r = io.compute_metrics(gt[None, None], torch.from_numpy(G["pred_lr"])[None, None], min_depth_eval=1e-3, max_depth_eval=80,
                           garg_crop=True, dataset="u4k")
    keys = [str(k) for k in G["metrics_resize_garg_keys"]]
    np.testing.assert_allclose([float(r[k]) for k in keys], G["metrics_resize_garg"], rtol=1e-6)
    assert "see" not in r
    r = io.compute_metrics(gt[None, None], pred[None, None], min_depth_eval=1e-3, max_depth_eval=80, disp_gt_edges=edges[None],
                           additional_mask=torch.from_numpy(G["additional_mask"])[None, None])
    keys = [str(k) for k in G["metrics_same_keys"]]
    np.testing.assert_allclose([float(r[k]) for k in keys], G["metrics_addmask"], rtol=1e-6)
    assert abs(G["metrics_addmask"] - G["metrics_same"]).max() > 0
