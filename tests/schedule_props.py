"""Size-independent properties of the tiled path, shared by the CPU wiring test (tiny config, torch stand-in ops) and
the GPU test at BASELINE.json's headline size (ViT-L, 2160x3840, 4x4 tiles, process_num 8), where the oracle is far
too slow to serve as the checker:
  * determinism: two passes give bit-identical maps;
  * schedule invariance: side-stream coarse branch + batches alternating over two streams == everything on one
    stream, bit-exactly (same kernels, only their interleaving changes);
  * batch invariance: halving process_num changes at most which kernel variant a layer runs on (tile shapes depend
    on the batch), so the maps agree within the stated compute-dtype tolerance;
  * the map is finite, has the reensemble shape and stays inside [min_depth, max_depth]."""
import torch


def check(m, lr, img, cfg, process_num, max_tol, mean_tol, cai_mode="m1"):
    def run(pn):
        d, _ = m(mode="infer", image_lr=lr, image_hr=img, cai_mode=cai_mode, process_num=pn)
        if d.is_cuda:
            torch.cuda.synchronize()
        return d.float().clone()

    saved = (m.overlap_coarse, m.overlap_batches, m.n_streams)
    d0 = run(process_num)
    assert torch.equal(d0, run(process_num)), "two passes differ"
    m.overlap_coarse, m.overlap_batches, m.n_streams = False, False, 1
    try:
        d1 = run(process_num)
    finally:
        m.overlap_coarse, m.overlap_batches, m.n_streams = saved
    assert torch.equal(d0, d1), f"stream schedule changes the result: max |diff| = {float((d0 - d1).abs().max())}"
    d2 = run(max(1, process_num // 2))
    diff = (d0 - d2).abs()
    assert float(diff.max()) <= max_tol and float(diff.mean()) <= mean_tol, (float(diff.max()), float(diff.mean()))
    ph, pw = cfg["patch_process_shape"]
    sh, sw = cfg["patch_split_num"]
    assert tuple(d0.shape) == (1, 1, ph * sh, pw * sw), tuple(d0.shape)
    assert bool(torch.isfinite(d0).all())
    assert float(d0.min()) >= float(cfg["min_depth"]) - 1e-6 and float(d0.max()) <= float(cfg["max_depth"]) + 1e-6
    return d0
