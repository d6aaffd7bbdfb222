"""GPU parity of the input / output side kernels (io.hip) against oracle/io_oracle.py and the
fixtures generated from the reference's own code (tests/golden/io_side.npz).  Tolerances: order statistics,
colour images and 16-bit export are integer / byte work -> bit-exact; the bicubic resize is evaluated in double on
both sides and compared after rounding to float32 (<= 1 ulp on at most 1e-5 of the pixels: summation order);
metrics are float32 terms summed in double here and pairwise in float32 by numpy -> rtol 2e-5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "io_side.npz"))


def _mods():
    from oracle import io_oracle
    from patchfusion_amd import postprocess as post
    from patchfusion_amd.hip_ops import ops
    from patchfusion_amd.preprocess import ImagePreprocessor
    return io_oracle, post, ops, ImagePreprocessor


def _close_f32(a, b, frac=1e-5):
    a, b = a.cpu().numpy(), np.asarray(b, dtype=np.float32)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    ulp[np.abs(a.astype(np.float64) - b.astype(np.float64)) <= 1e-9] = 0     # cancellation to ~0: ulps of a tiny value
    assert ulp.max() <= 1 and (ulp > 0).mean() <= frac, (ulp.max(), (ulp > 0).mean())


def test_preprocessor_vs_reference_fixture_and_torch_double():
    io, post, ops, Pre = _mods()
    r = Pre((96, 128), (28, 37))(G["img_u8"])
    _close_f32(r["image_hr"], G["read_image_96x128"].transpose(2, 0, 1))
    hr_ref, lr_ref = io.dataset_item(G["read_image_96x128"], (28, 37))
    assert float((r["image_lr"].cpu() - lr_ref).abs().max()) < 1e-5      # float32 source-index arithmetic of the bilinear resize
    r = Pre((61, 83), (28, 37))(G["img_u8"])                       # identity size: exact
    assert np.array_equal(r["image_hr"].cpu().numpy(), G["read_image_same"].transpose(2, 0, 1).astype(np.float32))
    # the reference's real geometry: 1080x1920 photo -> 2160x3840 (bicubic, double) and -> 392x518
    img = torch.randint(0, 256, (1080, 1920, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
    r = Pre((2160, 3840), (392, 518))(img)
    ref = io.read_image_arith(img.numpy(), (2160, 3840))
    _close_f32(r["image_hr"], ref.transpose(2, 0, 1))
    assert r["image_lr"].shape == (3, 392, 518)
    # 'u4k' raw files: no resize, channel reversal
    raw = torch.randint(0, 256, (270, 480, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(6))
    r = Pre((270, 480), (28, 37), dataset_name="u4k")(raw)
    assert np.array_equal(r["image_hr"].cpu().numpy(), io.read_image_arith(raw.numpy(), None, "u4k").transpose(2, 0, 1))


def test_percentiles_bit_exact():
    io, post, ops, _ = _mods()
    rs = np.random.RandomState(0)
    cases = {
        "normal_full": (rs.randn(1568 * 2072) * 3 + 5).astype(np.float32),
        "with_invalid": np.where(rs.rand(200000) < 0.1, -99, rs.rand(200000) * 80).astype(np.float32),
        "constant": np.full(100000, 0.7031, np.float32),
        "two_values": np.where(rs.rand(50001) < 0.97, 1.5, 2.5).astype(np.float32),
        "negatives": (-rs.rand(4097) * 10).astype(np.float32),
        "n1": np.array([3.25], np.float32),
        "n2": np.array([1.0, 2.0], np.float32),
        "denormal_and_zero": np.concatenate([np.zeros(100), -np.zeros(100), np.full(100, 1e-40)]).astype(np.float32),
    }
    for name, x in cases.items():
        for q0, q1 in ((2, 95), (0, 100), (50, 99.9)):
            inv = -99 if name == "with_invalid" else None
            got = ops.percentiles(torch.from_numpy(x).cuda(), q0, q1, invalid_val=inv).cpu().numpy()
            v = x[x != -99] if inv is not None else x
            ref = np.array([io.percentile_linear(v, q0), io.percentile_linear(v, q1)], np.float32)
            assert np.array_equal(got.view(np.int32), ref.view(np.int32)) or np.array_equal(got, ref), (name, q0, q1, got, ref)
    x = cases["normal_full"]
    got = ops.percentiles(torch.from_numpy(x).cuda(), 0, 100).cpu().numpy()
    assert got[0] == x.min() and got[1] == x.max()                 # size-independent property
    # all pixels invalid -> NaN (np.percentile of an empty selection)
    assert torch.isnan(ops.percentiles(torch.full((1000,), -99.0).cuda(), 2, 95, invalid_val=-99)).all()


def test_colorize_and_uint16_bit_exact():
    io, post, ops, _ = _mods()
    d = torch.from_numpy(G["depth"]).cuda()
    vmin, vmax = (float(np.float32(v)) for v in G["np_percentiles"])
    for cmap in ("magma_r", "gray_r"):
        img = post.colorize(d[None, None], vmin=vmin, vmax=vmax, cmap=cmap)        # reference fixture (its percentiles)
        assert np.array_equal(img.cpu().numpy(), G[f"colorize_{cmap}"])
        assert np.array_equal(post.colorize(d, cmap=cmap).cpu().numpy(), io.colorize(G["depth"], cmap=cmap))
    # stitched-map size, invalid pixels, constant map (vmin == vmax), out-of-range values
    rs = np.random.RandomState(3)
    big = (rs.rand(1568, 2072) ** 2 * 40).astype(np.float32)
    big[rs.rand(1568, 2072) < 0.02] = -99
    assert np.array_equal(post.colorize(torch.from_numpy(big).cuda(), cmap="magma_r").cpu().numpy(), io.colorize(big, cmap="magma_r"))
    const = np.full((64, 96), 0.7031, np.float32)
    assert np.array_equal(post.colorize(torch.from_numpy(const).cuda(), cmap="magma_r").cpu().numpy(), io.colorize(const, cmap="magma_r"))
    assert np.array_equal(post.colorize(torch.from_numpy(big).cuda(), vmin=5.0, vmax=20.0, cmap="gray_r").cpu().numpy(),
                          io.colorize(big, vmin=5.0, vmax=20.0, cmap="gray_r"))
    u = post.depth_to_uint16(torch.from_numpy(np.abs(G["depth"])).cuda()[None, None])
    assert np.array_equal(u.cpu().numpy(), G["uint16"])
    assert np.array_equal(post.depth_to_uint16(torch.from_numpy(np.abs(big)).cuda()).cpu().numpy(), io.depth_to_uint16(np.abs(big)))


def test_colorize_optional_arguments_bit_exact():
    """invalid_mask / gamma_corrected / value_transform (color.py:121-122, :86-91, :140-141) vs the reference-made fixtures and, at the
    stitched-map size with own percentiles, vs the oracle"""
    io, post, ops, _ = _mods()
    d, im = torch.from_numpy(G["depth"]).cuda()[None, None], G["invalid_mask"]
    lo, hi = (float(np.float32(v)) for v in G["np_percentiles"])
    lom, him = (float(np.float32(v)) for v in G["np_percentiles_mask"])
    assert np.array_equal(post.colorize(d, vmin=lom, vmax=him, cmap="magma_r", invalid_mask=im).cpu().numpy(), G["colorize_mask"])
    assert np.array_equal(post.colorize(d, vmin=lo, vmax=hi, cmap="magma_r", gamma_corrected=True).cpu().numpy(), G["colorize_gamma"])
    assert np.array_equal(post.colorize(d, vmin=lo, vmax=hi, cmap="gray_r", value_transform=np.square).cpu().numpy(), G["colorize_transform"])
    assert np.array_equal(post.colorize(d, vmin=lom, vmax=him, cmap="turbo_r", invalid_mask=torch.from_numpy(im).cuda(), gamma_corrected=True,
                                        value_transform=np.square, background_color=(10, 200, 30, 255)).cpu().numpy(), G["colorize_all"])
    rs = np.random.RandomState(4)
    big = (rs.rand(1568, 2072) ** 2 * 40).astype(np.float32)
    big[rs.rand(1568, 2072) < 0.02] = -99
    bm = rs.rand(1568, 2072) < 0.3
    for kw in (dict(invalid_mask=bm), dict(invalid_mask=bm, gamma_corrected=True, value_transform=np.sqrt), dict(gamma_corrected=True)):
        with np.errstate(invalid="ignore"):
            ref = io.colorize(big, cmap="magma_r", **kw)
            got = post.colorize(torch.from_numpy(big).cuda(), cmap="magma_r", **kw).cpu().numpy()
        assert np.array_equal(got, ref), sorted(kw)
    # masked percentiles alone (radix select over the mask's complement)
    p = ops.percentiles(torch.from_numpy(big).cuda(), 2, 95, invalid_val=-99, invalid_mask=torch.from_numpy(bm.astype(np.uint8)).cuda()).cpu().numpy()
    assert p[0] == io.percentile_linear(big[~bm], 2) and p[1] == io.percentile_linear(big[~bm], 95)


def test_metrics_vs_reference_fixture_and_oracle():
    io, post, ops, _ = _mods()
    gt, pred, edges = (torch.from_numpy(G[k]) for k in ("gt", "pred", "edges"))
    r = post.compute_metrics(gt.cuda()[None, None], pred.cuda()[None, None], min_depth_eval=1e-3, max_depth_eval=80, garg_crop=False,
                             eigen_crop=False, disp_gt_edges=edges[None])
    keys = [str(k) for k in G["metrics_same_keys"]]
    assert sorted(r) == keys
    np.testing.assert_allclose([r[k] for k in keys], G["metrics_same"], rtol=2e-5)
    r = post.compute_metrics(gt.cuda()[None, None], torch.from_numpy(G["pred_lr"]).cuda()[None, None], min_depth_eval=1e-3,
                             max_depth_eval=80, garg_crop=True, eigen_crop=False, dataset="u4k")
    keys = [str(k) for k in G["metrics_resize_garg_keys"]]
    np.testing.assert_allclose([r[k] for k in keys], G["metrics_resize_garg"], rtol=2e-5)
    r = post.compute_metrics(gt.cuda()[None, None], pred.cuda()[None, None], min_depth_eval=1e-3, max_depth_eval=80, garg_crop=False,
                             eigen_crop=False, disp_gt_edges=edges[None], additional_mask=torch.from_numpy(G["additional_mask"])[None, None])
    np.testing.assert_allclose([r[k] for k in [str(k) for k in G["metrics_same_keys"]]], G["metrics_addmask"], rtol=2e-5)
    # BASELINE geometry: 4K ground truth, stitched 1568x2072 prediction (resize inside the kernel), boundaries
    rs = np.random.RandomState(9)
    yy, xx = np.mgrid[0:540, 0:960].astype(np.float32)
    g = (5 + 3 * np.sin(xx / 40) * np.cos(yy / 30) + (xx > 480) * 4).astype(np.float32)
    g[:7] = 0
    p = torch.nn.functional.interpolate(torch.from_numpy(g)[None, None], (392, 518), mode="bilinear")[0, 0].numpy()
    p = (p * (1 + 0.03 * rs.randn(392, 518))).astype(np.float32)
    e = (np.abs(np.diff(g, axis=1, prepend=g[:, :1])) > 1).astype(np.float32)
    ref = io.compute_metrics(torch.from_numpy(g)[None, None], torch.from_numpy(p)[None, None], 1e-3, 80, disp_gt_edges=torch.from_numpy(e)[None])
    got = post.compute_metrics(torch.from_numpy(g).cuda(), torch.from_numpy(p).cuda(), min_depth_eval=1e-3, max_depth_eval=80,
                               garg_crop=False, eigen_crop=False, disp_gt_edges=torch.from_numpy(e))
    for k in ref:
        np.testing.assert_allclose(got[k], float(ref[k]), rtol=2e-5, err_msg=k)
    assert "libpf_hip.so" in open("/proc/self/maps").read()
