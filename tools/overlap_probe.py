"""GPU probe (round 5): can the HBM-bound Winograd transforms of one tile batch run under the matrix-bound split GEMM of the other?
Two streams run the SAME three-step Winograd layer (input transform -> batched persistent split GEMM -> output transform, csrc/winograd.hip run_split3)
on their own tensors, host issue interleaved A, B, A, B ... (what a layer-interleaved schedule of the two tile batches would issue).  Variants:
  grid   PF_W3_GRID: blocks of the persistent GEMM (256 = one per CU = today; fewer leave CUs to the other stream's transforms)
  token  PF_W3_TOKEN=1: the GEMMs of both streams are chained through one event (never two capped GEMMs at once)
  T      PF_W3_TGRID: the transforms as RESIDENT kernels on that many CUs (grid-stride) instead of one-shot launches that flood every free CU
Reported: ms per layer and stream (wall of 2 n layers / 2 n) against the single-stream time of the same layer (= fully serial).
usage: python tools/overlap_probe.py [n layers per stream, default 6]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchfusion_amd import hip_ops, packing as pk   # noqa: E402
from patchfusion_amd.hip_ops import ops             # noqa: E402

DEV = "cuda"


def run(layers_per_stream, streams, xs, ys, pw):
    """issue `layers_per_stream` layers on each stream, interleaved; -> ms per layer and stream"""
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for s in streams:
        s.wait_event(e0)
    for _ in range(layers_per_stream):
        for s, x, y in zip(streams, xs, ys):
            with torch.cuda.stream(s):
                ops.conv(x, pw, y, pad=1, act="relu")
    for s in streams:
        main.wait_stream(s)
    e1.record(main)
    e1.synchronize()
    return e0.elapsed_time(e1) / (layers_per_stream * len(streams))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    g = torch.Generator().manual_seed(0)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    # (GEMM grid, token, transform grid: 0 = one-shot transforms flooding the chip, n = resident on n CUs)
    variants = [(256, 0, 0), (192, 1, 64), (192, 1, 80), (176, 1, 80), (160, 1, 96), (208, 1, 48), (224, 1, 32), (192, 0, 64), (176, 0, 80), (192, 1, 0), (256, 0, 0)]
    print("| 3x3 layer @ B=8 | one stream ms/layer | one stream, transforms resident on 64 / 96 CUs | "
          + " | ".join(f"G {gr}{' +token' if tk else ''} T {tg or 'flood'}" for gr, tk, tg in variants) + " |")
    print("|---|---|---|" + "---|" * len(variants))
    for (cin, cout, H, W) in ((544, 544, 392, 518), (768, 768, 224, 296), (768, 256, 224, 296), (256, 256, 224, 296), (768, 768, 112, 148)):
        w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(cout), dtype=torch.float32).to(DEV)
        xs = [torch.randn(8, H, W, cin, device=DEV) for _ in range(2)]
        ys = [torch.empty(8, H, W, cout, device=DEV) for _ in range(2)]
        for k in ("PF_W3_GRID", "PF_W3_TOKEN", "PF_W3_TGRID"):
            os.environ.pop(k, None)
        hip_ops.refresh_env()
        run(2, [sa, sb], xs, ys, pw)                      # warm-up (workspaces of both streams)
        one = run(n, [sa], xs[:1], ys[:1], pw)
        os.environ["PF_W3_TGRID"] = "64"
        one64 = run(n, [sa], xs[:1], ys[:1], pw)
        os.environ["PF_W3_TGRID"] = "96"
        one96 = run(n, [sa], xs[:1], ys[:1], pw)
        cells = []
        for gr, tk, tg in variants:
            os.environ["PF_W3_GRID"] = str(gr)
            os.environ["PF_W3_TOKEN"] = str(tk)
            os.environ["PF_W3_TGRID"] = str(tg)
            run(1, [sa, sb], xs, ys, pw)
            cells.append(run(n, [sa, sb], xs, ys, pw))
        print(f"| {cin}->{cout} @ {H}x{W} | {one:.3f} | {one64:.3f} / {one96:.3f} | " + " | ".join(f"{c:.3f}" for c in cells) + " |", flush=True)
        del xs, ys, pw
        torch.cuda.empty_cache()
    for k in ("PF_W3_GRID", "PF_W3_TOKEN", "PF_W3_TGRID"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
