"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/io_side.npz by running the REFERENCE'S OWN PYTHON for the
input / output side of the path on small seeded inputs:

  read_image()      estimator/datasets/general_dataset.py:22-47  (cv2.imread / cvtColor are stubbed to hand it a
                    seeded uint8 image - the decode itself is out of scope, the arithmetic after it is not)
  colorize()        estimator/utils/color.py:95-150              (cmap magma_r and gray_r, as tester.py:68-71)
  compute_metrics() estimator/utils/metric.py:87-148             (with and without resize / edges)
  uint16 export     estimator/tester/tester.py:75                (the expression itself)

Run in the build container only:   python -m oracle.make_golden_io
NOTE: executes under THIS image's numpy 2.2 / matplotlib 3.10 (the reference pins 1.24.4 / 3.7.3): the percentile
inside colorize() is the installed numpy's.  The fixture records it (vmin/vmax) so tests can separate the two effects.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "io_side.npz")


def synth_depth(h, w, seed, invalid_frac=0.0):
    g = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    d = 2.0 + 1.5 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + 0.002 * xx + g.rand(h, w).astype(np.float32) * 0.3
    d[h // 3:h // 2, w // 4:w // 2] += 3.0                      # a depth step -> edges
    if invalid_frac:
        d[g.rand(h, w) < invalid_frac] = -99
    return d.astype(np.float32)


def main():
    ref_shim.import_reference()
    import cv2  # the stub module
    from estimator.datasets.general_dataset import read_image
    from estimator.utils import colorize
    from estimator.utils.metric import compute_metrics, get_boundaries

    out = {}
    # ---- read_image: 61x83 uint8 -> bicubic 96x128 and identity size ----
    rgb = np.random.RandomState(3).randint(0, 256, (61, 83, 3), dtype=np.uint8)
    cv2.imread = lambda path: rgb[:, :, ::-1].copy()            # "file" in BGR like OpenCV
    cv2.COLOR_BGR2RGB, cv2.COLOR_GRAY2BGR = 4, 8
    cv2.cvtColor = lambda img, code: img[:, :, ::-1] if code == 4 else np.repeat(img[:, :, None], 3, 2)
    out["img_u8"] = rgb
    out["read_image_96x128"] = read_image("x.png", "general", (96, 128)).astype(np.float64)
    out["read_image_same"] = read_image("x.png", "general", (61, 83)).astype(np.float64)

    # ---- colorize ----
    d = synth_depth(120, 164, 5, invalid_frac=0.01)
    out["depth"] = d
    for cmap in ("magma_r", "gray_r"):
        out[f"colorize_{cmap}"] = colorize(torch.from_numpy(d)[None, None], cmap=cmap)
    # the optional arguments (color.py:121-122 invalid_mask, :140-141 value_transform, :86-91 gamma_corrected)
    im = np.zeros(d.shape, bool)
    im[20:50, 30:90] = True
    im[np.random.RandomState(6).rand(*d.shape) < 0.02] = True
    out["invalid_mask"] = im
    out["np_percentiles_mask"] = np.array([np.percentile(d[~im], 2), np.percentile(d[~im], 95)], dtype=np.float64)
    out["colorize_mask"] = colorize(torch.from_numpy(d)[None, None], cmap="magma_r", invalid_mask=im.copy())
    out["colorize_gamma"] = colorize(torch.from_numpy(d)[None, None], cmap="magma_r", gamma_corrected=True)
    out["colorize_transform"] = colorize(torch.from_numpy(d)[None, None], cmap="gray_r", value_transform=np.square)
    out["colorize_all"] = colorize(torch.from_numpy(d)[None, None], cmap="turbo_r", invalid_mask=im.copy(), gamma_corrected=True,
                                   value_transform=np.square, background_color=(10, 200, 30, 255))
    m = d != -99
    out["np_percentiles"] = np.array([np.percentile(d[m], 2), np.percentile(d[m], 95)], dtype=np.float64)
    out["uint16"] = (torch.from_numpy(np.abs(d))[None, None].clone().squeeze().detach().cpu().numpy() * 256).astype("uint16")

    # ---- compute_metrics: same grid + edges, and pred on a coarser grid (resize) ----
    gt = synth_depth(120, 164, 7)
    gt[:4] = 0.0                                                  # invalid gt rows (below min depth)
    noise = np.random.RandomState(8).randn(120, 164).astype(np.float32)
    pred = gt * (1.0 + 0.05 * noise) + 0.02
    pred[10, 10], pred[11, 11], pred[12, 12] = np.inf, np.nan, -1.0
    cv2.dilate = lambda e, k, iterations=1: e                    # get_boundaries: dilation stub = identity
    edges = get_boundaries(gt, th=0.5, dilation=0)
    out["gt"], out["pred"], out["edges"] = gt, pred, edges
    r = compute_metrics(torch.from_numpy(gt)[None, None], torch.from_numpy(pred.copy())[None, None], disp_gt_edges=torch.from_numpy(edges)[None],
                        min_depth_eval=1e-3, max_depth_eval=80, garg_crop=False, eigen_crop=False, dataset="u4k")
    out["metrics_same_keys"] = np.array(sorted(r.keys()))
    out["metrics_same"] = np.array([float(r[k]) for k in sorted(r.keys())], dtype=np.float64)
    pred_lr = torch.nn.functional.interpolate(torch.from_numpy(pred.copy())[None, None].nan_to_num(1.0, 1.0, 1.0), (60, 82), mode="bilinear").squeeze().numpy()
    out["pred_lr"] = pred_lr
    r = compute_metrics(torch.from_numpy(gt)[None, None], torch.from_numpy(pred_lr.copy())[None, None], min_depth_eval=1e-3, max_depth_eval=80,
                        garg_crop=True, eigen_crop=False, dataset="u4k")
    out["metrics_resize_garg"] = np.array([float(r[k]) for k in sorted(r.keys())], dtype=np.float64)
    out["metrics_resize_garg_keys"] = np.array(sorted(r.keys()))
    am = np.random.RandomState(9).rand(120, 164) < 0.6            # additional_mask (metric.py:128-130)
    out["additional_mask"] = am
    r = compute_metrics(torch.from_numpy(gt)[None, None], torch.from_numpy(pred.copy())[None, None], disp_gt_edges=torch.from_numpy(edges)[None],
                        min_depth_eval=1e-3, max_depth_eval=80, garg_crop=False, eigen_crop=False, dataset="u4k",
                        additional_mask=torch.from_numpy(am)[None, None])
    out["metrics_addmask"] = np.array([float(r[k]) for k in sorted(r.keys())], dtype=np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
