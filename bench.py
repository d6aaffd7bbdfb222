"""Headline benchmark: PatchFusion tiled inference, Depth-Anything ViT-L, 4K image, P=16 tiles
(BASELINE.json configs[2]) on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (coarse branch + G2L + per-tile fine branch + guided fusion + stitch)
over one synthetic 2160x3840 image that is already resident in HBM.  Weak scaling: every GPU processes 16
tiles per step -- at N GPUs the image is split into 16*N tiles (4x4, 4x8, 8x8, 8x16) that are sharded over
the ranks (coarse branch + G2L replicated, no communication) and the per-tile depths are all-gathered over
RCCL before every rank stitches.  value = tiles processed by all ranks / max-over-ranks wall time.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     : the dominant kernel (implicit-GEMM 3x3 conv 544->544 @ 8x392x518, the largest single op of
                 the fusion U-Net) timed live with HIP events on its launch stream, vs the dense MFMA peak
  cpu_baseline : the oracle (CPU port of the reference, oracle/pf_oracle.py) timed on the host cores for a
                 bounded sample (one tile = fine branch + fusion) of the same workload  [N=1, rank 0 only]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPLITS = {1: (4, 4), 2: (4, 8), 4: (8, 8), 8: (8, 16)}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # MI355X_MICROARCH.md: dense MFMA peaks


T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default=os.environ.get("PF_BENCH_DTYPE", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--process-num", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only time the dominant kernel (kernel tuning aid)")
    ap.add_argument("--gemm-sweep", action="store_true", help="time the ViT-L linear layers / other shapes (kernel tuning aid)")
    args = ap.parse_args()

    if args.gemm_sweep:
        torch.cuda.set_device(0)
        gemm_sweep(args.dtype, torch.device("cuda", 0))
        return
    if args.roofline_only:
        torch.cuda.set_device(0)
        print(json.dumps(roofline(args.dtype, torch.device("cuda", 0))), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    N = max(world, 1)
    assert args.gpus == N or world == 1, "launch with torch.distributed.run for --gpus > 1"

    from patchfusion_amd.config import make_config
    from patchfusion_amd.model import PatchFusion
    from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict

    raw = (2160, 3840)
    split = SPLITS.get(N, (4, 4))
    cfg = make_config(args.encoder, (392, 518), raw, split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    model = PatchFusion(cfg, compute_dtype=args.dtype).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    log("model built and on device")
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234)).to(dev)
    lr = model.resizer(img)
    P = split[0] * split[1]

    def step():
        d, _ = model(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=args.process_num)
        return d

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    log(f"timed {args.steps} steps: {dt:.3f}s")
    ms = dt / args.steps * 1e3
    value = P * args.steps / dt

    out = {
        "metric": "patches/sec (DepthAnything-ViT-L PatchFusion, 4K image, P=16 tiles per GPU)",
        "value": round(value, 3), "unit": "patches/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "ms_per_4k_image": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"Depth-Anything-{args.encoder}14 PatchFusion, 2160x3840 synthetic RGB, "
                               f"{split[0]}x{split[1]} regular tiling (cai_mode m1, {P} tiles = 16 per GPU), process_num={args.process_num}, "
                               "random-init weights (BASELINE.json configs[2] at N=1)",
                   "parallelism": f"patch-sharded x{N}, coarse+G2L replicated, RCCL all_gather of tile depths" if N > 1 else "single GPU"},
    }

    if rank == 0 and not args.no_roofline:
        out["roofline"] = roofline(args.dtype, dev)
        log(f"roofline: {out['roofline']}")
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, img.cpu())
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()                      # the other ranks wait for rank 0's roofline launch, then all leave together
        torch.distributed.destroy_process_group()


def gemm_sweep(dtype, dev):
    from patchfusion_amd import packing as pk
    from patchfusion_amd.hip_ops import ops
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    shapes = [("qkv", 8296, 1024, 3072, 1), ("proj", 8296, 1024, 1024, 1), ("fc1", 8296, 1024, 4096, 1), ("fc2", 8296, 4096, 1024, 1),
              ("up4_1", 8 * 224 * 296, 768, 768, 3), ("up4_2", 8 * 224 * 296, 768, 256, 3), ("c544_32", 8 * 392 * 518, 544, 32, 3),
              ("c64_32", 8 * 392 * 518, 64, 32, 3), ("c256_256_L4", 8 * 224 * 296, 512, 256, 3),
              ("up3_768_L3", 8 * 112 * 148, 768, 768, 3), ("up2_768_L2", 8 * 56 * 74, 768, 768, 3), ("up1_768_L1", 8 * 28 * 37, 768, 768, 3),
              ("c512_256_L2", 8 * 56 * 74, 512, 256, 3), ("c512_256_L1", 8 * 28 * 37, 512, 256, 3), ("c512_256_L0", 8 * 14 * 19, 512, 256, 3),
              ("rcu256_L3", 8 * 112 * 148, 256, 256, 3)]
    for name, M, K, N, k in shapes:
        if k == 1:
            x = torch.randn(1, 1, M, K, device=dev).to(tdt)
            y = torch.empty(1, 1, M, N, device=dev, dtype=tdt)
            w = torch.randn(N, K) / K ** 0.5
        else:
            hw = {8 * 224 * 296: (224, 296), 8 * 392 * 518: (392, 518), 8 * 112 * 148: (112, 148), 8 * 56 * 74: (56, 74),
                  8 * 28 * 37: (28, 37), 8 * 14 * 19: (14, 19)}[M]
            x = torch.randn(8, hw[0], hw[1], K, device=dev).to(tdt)
            y = torch.empty(8, hw[0], hw[1], N, device=dev, dtype=tdt)
            w = torch.randn(N, K, 3, 3) / (9 * K) ** 0.5
        pw = pk.pack_conv(w, torch.zeros(N), dtype=tdt).to(dev)
        ms = ops.conv(x, pw, y, pad=k // 2, _timed=5)
        fl = 2.0 * M * N * K * k * k
        print(f"{name:12s} M={M} K={K}x{k}x{k} N={N}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
        del x, y


def roofline(dtype, dev):
    """Dominant kernel: conv_igemm_kernel on the fusion U-Net's 3x3 conv 544->544 @ [8,392,518] (vitl):
    algorithmic FLOPs = 2 * 8*392*518 * 9*544 * 544 per launch."""
    from patchfusion_amd import packing as pk
    from patchfusion_amd.hip_ops import ops
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    B, H, W, C = 8, 392, 518, 544
    w = torch.randn(C, C, 3, 3) / (9 * C) ** 0.5
    pw = pk.pack_conv(w, torch.zeros(C), dtype=tdt).to(dev)
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    y = torch.empty(B, H, W, C, device=dev, dtype=tdt)
    ms = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
    flops = 2.0 * B * H * W * 9 * C * C
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    traffic = None   # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside bench.py): profiles/r1_pmc_dominant_kernel.json
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_dominant_kernel.json")) as f:
            traffic = json.load(f)["derived"]["traffic_bytes"] if dtype == "bf16" else None
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "conv3x3_halo_kernel<2,4,3> (bf16) / conv_igemm_kernel (fp32): 3x3 544->544 @ 8x392x518, the GuidedFusion up-conv = largest op",
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic}


def cpu_baseline(cfg, sd, img):
    """The oracle (CPU restatement of the reference, kind 'port') on the host cores, bounded sample: ONE tile of
    the same workload = fine branch + fusion_forward (4029.9 GFLOP, G2L hoisted like the engine does); the coarse
    pass + G2L it needs are computed untimed.  Threads are capped at 64: with all 256 hardware threads of the
    GPU box torch's CPU kernels oversubscribe (measured 108 s instead of ~5 s for one branch forward)."""
    from oracle import pf_oracle
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    orc = pf_oracle.Oracle(cfg, sd)
    with torch.no_grad():
        lr = orc.resizer(img)
        orc.coarse_depth, orc.coarse_feats = pf_oracle.branch_forward(sd, "coarse_branch.", lr, cfg["coarse_branch"])
        orc.g2l = pf_oracle.g2l_all(sd, orc.coarse_feats)
        tile_cfg = pf_oracle.prepare_tile_cfg(orc.ps, cfg["image_raw_shape"], cfg["patch_split_num"])
        hr, wr = tile_cfg["patch_raw_shape"]
        crop = orc.resizer(img[:, :, :hr, :wr])
        box = torch.tensor([[0, 0, wr, hr]]).int()
        log("cpu baseline: coarse pass done, timing one tile")
        t0 = time.perf_counter()
        orc._predict(crop, box, tile_cfg, 1)
        dt = time.perf_counter() - t0
    log(f"cpu baseline: one tile (fine branch + fusion) {dt:.1f}s on {cores} threads")
    return {"value": round(1.0 / dt, 5), "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"1 tile (fine branch + fusion, 4029.9 GFLOP, G2L hoisted) of the same DA-vitl 392x518 workload: "
                      f"{dt:.1f} s on {cores} threads"}


if __name__ == "__main__":
    main()
