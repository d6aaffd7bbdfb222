"""Headline benchmark: PatchFusion tiled inference, Depth-Anything ViT-L, 4K image, P=16 tiles
(BASELINE.json configs[2]) on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (coarse branch + G2L + per-tile fine branch + guided fusion + stitch)
over one synthetic 2160x3840 image that is already resident in HBM.

HEADLINE PRECISION = the reference's: float32 end to end (`dtype: "f32"`, exact mode: f32 storage and float32-grade arithmetic --
the f32 MFMA v_mfma_f32_16x16x4_f32, and for the large GEMMs (ViT block linears, attention, the transform-domain GEMM of the widest
Winograd layers) split-precision products on the bf16 MFMA with f32 accumulation whose error against float64 is measured <= the f32
MFMA kernels'; `config.precision` says so and the object `f32_mfma_only` carries the same pass with every GEMM on the f32 MFMA; parity
bar 1e-4 in depth units at this configuration, 2e-4 against the reference's own outputs).
The fast mode (bf16 storage / bf16 MFMA, f32 accumulation and f32 metric-bins head) is timed on the same image with
the same K/W and reported in the secondary object `"bf16"` TOGETHER WITH ITS MEASURED ERROR against the f32 map of
the same run (max / p99 / mean |delta| in depth units) -- it is never the headline `value`.

Tiles per step: N=1 4x4 (configs[2], 16 tiles), N=2 4x8 (32), N=4 8x8 (64) -- 16 tiles per GPU, weak scaling --
and N=8 8x8 = 64 tiles sharded 8-way, 8 per GPU (BASELINE.json configs[3]).  The tiles of the image are sharded over
the ranks (explicit opt-in shard_patches=True; coarse branch + G2L replicated, no communication) and the per-tile
depths are all-gathered over RCCL before every rank stitches.  value = tiles of the image / max-over-ranks wall time.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     : the dominant launch of the headline mode (of the 3x3 conv 544->544 @ 8x392x518, the largest single layer of the
                 fusion U-Net: in the default dispatch the batched split-precision GEMM over its 36 Winograd transform points) timed
                 live with HIP events on its launch stream: USEFUL float32 FLOPs vs the f32 MFMA peak (+ executed bf16 FLOPs vs the bf16 peak)
  f32_mfma_only: {value, ms_per_step, ...} the same pass with PF_LINEAR_SPLIT3=0 PF_WINO_SPLIT3=0 (N=1 only)
  bf16         : {value, ms_per_step, err{...}, roofline{...}}  (N=1 only)
  cpu_baseline : kind "reference" = the reference's own PatchFusion (oracle/ref_shim.py) when /root/reference exists
                 on the box, else kind "port" = the oracle (oracle/pf_oracle.py); bounded sample, N=1 rank 0 only
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPLITS = {1: (4, 4), 2: (4, 8), 4: (8, 8), 8: (8, 8)}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # MI355X_MICROARCH.md: dense MFMA peaks
PRACTICAL_BF16_TFLOPS = 2020.0                     # measured on this chip, random h / m / l operands, register-resident v_mfma_f32_16x16x32_bf16 (profiles/r5_mfma_ceiling.md)
DTYPE_NAME = {"fp32": "f32", "bf16": "bf16"}

T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default=os.environ.get("PF_BENCH_DTYPE", "fp32"), choices=["bf16", "fp32"],
                    help="headline mode (default fp32 = the reference's precision)")
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--process-num", type=int, default=8)
    ap.add_argument("--split", default=None, help="override the tile grid, e.g. 8x16 (kernel / scaling experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary bf16 leg")
    ap.add_argument("--roofline-only", action="store_true", help="only time the dominant kernel (kernel tuning aid)")
    ap.add_argument("--gemm-sweep", action="store_true", help="time the ViT-L linear layers / other shapes (kernel tuning aid)")
    ap.add_argument("--only", default="", help="gemm-sweep: comma separated shape names to run")
    ap.add_argument("--dry-run", action="store_true", default=os.environ.get("PF_BENCH_DRY_RUN", "0") == "1",
                    help="the launch line, rendezvous, tile sharding, all-gather, timing protocol and JSON line of a --gpus N run WITHOUT GPUs: gloo backend, "
                         "CPU tensors, the torch stand-in op set of the CPU tests (tests/fake_ops.py) on a tiny ViT-S geometry with the SAME tile grid.  "
                         "Its numbers mean nothing (`dry_run: true`); tests/test_bench_dry_run_cpu.py runs the driver's 8-rank command through it")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    if args.gemm_sweep:
        torch.cuda.set_device(0)
        gemm_sweep(args.dtype, torch.device("cuda", 0), [v for v in args.only.split(",") if v])
        return
    if args.roofline_only:
        torch.cuda.set_device(0)
        print(json.dumps(roofline(args.dtype, torch.device("cuda", 0), gemm_only=True)), flush=True)   # one kernel only: PMC passes
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
    split = tuple(int(v) for v in args.split.split("x")) if args.split else SPLITS.get(N, (4, 4))
    cfg = make_config(args.encoder, (392, 518), raw, split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234)).to(dev)
    P = split[0] * split[1]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    mem, rank_s = {}, {}

    def run_mode(dtype):
        """W untimed + K timed steps of the whole hot path in `dtype`; -> (seconds for K steps, last depth map)."""
        model = PatchFusion(cfg, compute_dtype=dtype, shard_patches=world > 1).eval()
        model.load_state_dict(sd, strict=True)
        model = model.to(dev)
        lr = model.resizer(img)
        torch.cuda.reset_peak_memory_stats(dev)

        def step():
            d, _ = model(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=args.process_num)
            return d

        d = None
        for i in range(args.warmup):
            d = step()
            torch.cuda.synchronize()
            log(f"{dtype}: warmup step {i} done")
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d = step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(t) for _ in range(world)]
            torch.distributed.all_gather(every, t)           # per-rank wall times: a straggler GPU is visible in the line
            rank_s[dtype] = [round(float(v.item()), 4) for v in every]
            dt = max(rank_s[dtype])
        log(f"{dtype}: timed {args.steps} steps: {dt:.3f}s")
        mem[dtype] = {"max_allocated_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                      "max_reserved_GB": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2)}
        d = d.float().clone()
        del model
        torch.cuda.empty_cache()
        return dt, d

    dt, depth = run_mode(args.dtype)
    ms = dt / args.steps * 1e3
    value = P * args.steps / dt
    per_gpu = P // N
    if N <= 4:
        scaling, which = "weak", f"BASELINE.json configs[2] at N=1; {per_gpu} tiles per GPU"
    else:
        scaling, which = "strong", "BASELINE.json configs[3]: P=64 tiles sharded 8-way, 8 per GPU (same image and tile count as N=4)"
    out = {
        "metric": "patches/sec (DepthAnything-ViT-L PatchFusion, 4K image, P=16 tiles at N=1)",
        "value": round(value, 3), "unit": "patches/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "ms_per_4k_image": round(ms, 3), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": DTYPE_NAME[args.dtype], "data": "synthetic",
        "config": {"workload": f"Depth-Anything-{args.encoder}14 PatchFusion, 2160x3840 synthetic RGB, "
                               f"{split[0]}x{split[1]} regular tiling (cai_mode m1, {P} tiles), process_num={args.process_num}, "
                               f"random-init weights ({which})",
                   "precision": PRECISION_F32[split3_on()] if args.dtype == "fp32"
                                else "bf16 storage + bf16 MFMA, f32 accumulation, f32 metric-bins head",
                   "parallelism": f"patch-sharded x{N}, coarse+G2L replicated, RCCL all_gather of tile depths" if N > 1 else "single GPU"},
    }

    out["memory"] = dict(mem[args.dtype], note="torch allocator peak on rank 0 over warm-up + timed steps: weights (two f32 checkpoints + packed "
                         "copies), activations of process_num tiles on two streams, the three-step Winograd V/M arenas (one pair per stream, capped at "
                         "PF_WS_CAP_GB = 2.5 GB: larger layers run as sub-batches of their tiles); PatchFusion.free_parameters() (opt-in) returns another ~2.7 GB")
    if world > 1:
        out["rank_seconds"] = rank_s[args.dtype]
        # self-diagnosis of the first real multi-GPU run (round-5 review item 8): the communicator's own size after an actual collective on the device, and
        # the tile-depth gather of one image pass on its own -- [P, 392, 518] float32 through patchfusion_amd.dist.all_gather_shards (one
        # all_gather_into_tensor over RCCL), timed with a barrier on both sides, maximum over ranks, after one untimed call (RCCL sets its rings up lazily)
        out["rccl"] = gather_diagnosis(P, (392, 518), world, dev, barrier, ms)
    if rank == 0 and not args.no_roofline:
        out["roofline"] = roofline(args.dtype, dev)
        if args.dtype == "fp32" and split3_on():
            out["roofline"]["attention"] = roofline_attention(dev)       # the second north-star kernel beside the dominant launch (round-5 review item 7)
        log(f"roofline: {out['roofline']}")
    if N == 1 and args.dtype == "fp32" and not args.no_secondary and split3_on():
        # the same pass with every GEMM on the float32 MFMA (PF_LINEAR_SPLIT3=0 PF_WINO_SPLIT3=0): the number to quote if split-precision linears are not
        # accepted as "float32", and the measured distance between the two float32 evaluations
        os.environ["PF_LINEAR_SPLIT3"] = "0"
        os.environ["PF_WINO_SPLIT3"] = "0"
        try:
            dt3, depth3 = run_mode("fp32")
        finally:
            os.environ.pop("PF_LINEAR_SPLIT3", None)
            os.environ.pop("PF_WINO_SPLIT3", None)
        d3 = (depth3 - depth).abs()
        out["f32_mfma_only"] = {"value": round(P * args.steps / dt3, 3), "unit": "patches/s", "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                                "precision": PRECISION_F32[False],
                                "max_abs_depth_diff_vs_headline": float(d3.max()), "mean_abs_depth_diff_vs_headline": float(d3.mean())}
        log(f"f32 MFMA only: {out['f32_mfma_only']}")
        del depth3
    if N == 1 and args.dtype == "fp32" and not args.no_secondary:
        dt2, depth2 = run_mode("bf16")
        diff = (depth2 - depth).abs().flatten()
        sec = {"value": round(P * args.steps / dt2, 3), "unit": "patches/s", "ms_per_step": round(dt2 / args.steps * 1e3, 3),
               "dtype": "bf16", "precision": "bf16 storage + bf16 MFMA, f32 accumulation, f32 metric-bins head",
               "err_vs_f32_same_run": {"max_abs": float(diff.max()), "p99_abs": float(torch.quantile(diff[::7], 0.99)),
                                       "mean_abs": float(diff.mean()), "depth_std": float(depth.std()),
                                       "depth_min": float(depth.min()), "depth_max": float(depth.max()), "unit": "depth units"}}
        if not args.no_roofline:
            sec["roofline"] = roofline("bf16", dev)
        out["bf16"] = sec
        log(f"bf16 secondary: {sec}")
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, img.cpu())
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()                      # the other ranks wait for rank 0's roofline launch, then all leave together
        torch.distributed.destroy_process_group()


def gather_diagnosis(P, tile_hw, world, dev, barrier, ms_per_step):
    """The communicator's size after an actual collective and the tile-depth gather of one image pass on its own: [P, h, w] float32 through
    patchfusion_amd.dist.all_gather_shards (one all_gather_into_tensor), a barrier on both sides, maximum over ranks, after one untimed call."""
    import torch.distributed as dist
    from patchfusion_amd.dist import all_gather_shards
    from patchfusion_amd.tiling import shard_range
    tiles = torch.zeros(P, *tile_hw, device=dev)
    lo, hi = shard_range(P, dist.get_rank(), world)
    tiles[lo:hi] = dist.get_rank() + 1.0
    g_out = all_gather_shards(tiles, P, world)
    ok = all(bool((g_out[slice(*shard_range(P, r, world))] == r + 1.0).all()) for r in range(world))      # every rank's rows arrived in tile order
    barrier()
    t0 = time.perf_counter()
    n_g = 10
    for _ in range(n_g):
        g_out = all_gather_shards(tiles, P, world)
    barrier()
    tg = torch.tensor([(time.perf_counter() - t0) / n_g * 1e3], device=dev, dtype=torch.float64)
    dist.all_reduce(tg, op=dist.ReduceOp.MAX)
    return {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "tile_gather_ms": round(float(tg.item()), 4), "rows_in_tile_order": ok,
            "tile_gather_bytes_per_rank": int(tiles[0].numel() * 4 * (hi - lo)), "gathered_shape": list(g_out.shape),
            "share_of_ms_per_step": round(float(tg.item()) / ms_per_step, 4)}


def dry_run(args):
    """see --dry-run: every line of the multi-rank protocol of main() (env rendezvous, shard_patches model, barrier + max-over-ranks timing, per-rank
    seconds, rank 0 prints ONE JSON line, common exit) on CPU tensors."""
    import torch.distributed as dist
    from patchfusion_amd.config import make_config
    from patchfusion_amd.model import PatchFusion
    from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict
    from tests.fake_ops import ops as fake_ops
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo")
    N = max(world, 1)
    assert args.gpus == N or world == 1, "launch with torch.distributed.run for --gpus > 1"
    split = tuple(int(v) for v in args.split.split("x")) if args.split else SPLITS.get(N, (4, 4))
    ps = (112, 154)
    raw = (ps[0] * split[0], ps[1] * split[1])          # the same tile GRID as the real run on a tiny geometry
    cfg = make_config("vits", ps, raw, split)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    img = torch.rand(1, 3, *raw, generator=torch.Generator().manual_seed(1234))
    P = split[0] * split[1]
    model = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops, shard_patches=world > 1).eval()
    model.load_state_dict(sd, strict=True)
    lr = model.resizer(img)

    def barrier():
        if world > 1:
            dist.barrier()

    d = None
    for _ in range(args.warmup):
        d, _x = model(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=args.process_num)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d, _x = model(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=args.process_num)
    barrier()
    dt = time.perf_counter() - t0
    rank_s = [round(dt, 4)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_s = [round(float(v.item()), 4) for v in every]
        dt = max(rank_s)
        # every rank holds the SAME stitched map (the gather is an all-gather): checksum over the ranks
        c = torch.tensor([float(d.double().sum())], dtype=torch.float64)
        cs = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(cs, c)
        assert all(float(v) == float(cs[0]) for v in cs), "ranks disagree on the stitched map"
    comm = gather_diagnosis(P, ps, world, torch.device("cpu"), barrier, dt / args.steps * 1e3) if world > 1 else None
    if rank == 0:
        print(json.dumps({
            "metric": "patches/sec (DRY RUN of the multi-rank protocol on CPU tensors -- not a measurement)", "value": round(P * args.steps / dt, 3),
            "unit": "patches/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if N <= 4 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dry_run": True, "rank_seconds": rank_s, "depth_shape": list(d.shape), "rccl": comm,
            "config": {"workload": f"DRY RUN: Depth-Anything-vits14 geometry {ps[0]}x{ps[1]}, {raw[0]}x{raw[1]} image, {split[0]}x{split[1]} tiles "
                                   f"({P} tiles, {P // N} per rank), gloo, torch stand-in ops",
                       "parallelism": f"patch-sharded x{N}, coarse+G2L replicated, gloo all_gather of tile depths" if N > 1 else "single process"}}), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


PRECISION_F32 = {
    True: "float32 storage; convolutions on the f32 MFMA, except the GEMM families that run in split precision: the ViT block linears, the ViT "
          "attention (QK^T and PV) and the transform-domain GEMM of the three-step Winograd layers (each f32 operand = three bf16 planes, six "
          "partial products on the bf16 MFMA, f32 accumulation: error vs float64 <= the f32 MFMA kernel's, tests/op_checks.py gemm_split3; "
          "PF_LINEAR_SPLIT3=0 PF_WINO_SPLIT3=0 -> f32_mfma_only)",
    False: "float32 storage + f32 MFMA everywhere (exact mode = the reference's precision)"}


def split3_on():
    from patchfusion_amd.engine import linear_split3_enabled
    return linear_split3_enabled()


def gemm_sweep(dtype, dev, only=()):
    from patchfusion_amd import packing as pk
    from patchfusion_amd.hip_ops import ops
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    shapes = [("qkv", 8296, 1024, 3072, 1), ("proj", 8296, 1024, 1024, 1), ("fc1", 8296, 1024, 4096, 1), ("fc2", 8296, 4096, 1024, 1),
              ("c544_544", 8 * 392 * 518, 544, 544, 3),
              ("up4_1", 8 * 224 * 296, 768, 768, 3), ("up4_2", 8 * 224 * 296, 768, 256, 3), ("c544_32", 8 * 392 * 518, 544, 32, 3),
              ("c64_32", 8 * 392 * 518, 64, 32, 3), ("c256_256_L4", 8 * 224 * 296, 512, 256, 3),
              ("up3_768_L3", 8 * 112 * 148, 768, 768, 3), ("up2_768_L2", 8 * 56 * 74, 768, 768, 3), ("up1_768_L1", 8 * 28 * 37, 768, 768, 3),
              ("c512_256_L2", 8 * 56 * 74, 512, 256, 3), ("c512_256_L1", 8 * 28 * 37, 512, 256, 3), ("c512_256_L0", 8 * 14 * 19, 512, 256, 3),
              ("rcu256_L3", 8 * 112 * 148, 256, 256, 3)]
    for name, M, K, N, k in shapes:
        if only and name not in only:
            continue
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


def kernel_source_sha():
    h = hashlib.sha256()
    for f in ("igemm.hip", "wino_fused.hip", "gemm_split3.hip", "winograd.hip", "pf_common.h"):
        with open(os.path.join(ROOT, "patchfusion_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def roofline_attention(dev):
    """The second north-star kernel, timed live: the ViT split attention launch of the fine branch (pf_vit_attention_split3_v2, csrc/attn_split3.hip) at the
    pass's shape -- 8 tiles x 16 heads x 1037 tokens, head_dim 64, real h / m / l planes of random float32 q / k / v in, chunk-major planes out.  Same accounting
    as the dominant launch: USEFUL float32 FLOPs 4 B H S^2 64 (QK^T and PV, padding not counted) against 2500 / 6 TF/s (six bf16 MFMAs per product)."""
    from patchfusion_amd.hip_ops import ops
    B, S, Hh = 8, 1037, 16
    D = Hh * 64
    qkv = torch.randn(B * S, 3 * D, device=dev) * 0.5
    q3 = torch.empty(3, B * S, 3 * D, dtype=torch.bfloat16, device=dev)
    ops.split3(qkv, q3)
    out = torch.empty(3, D // 32, B * S, 32, dtype=torch.bfloat16, device=dev)
    # three rounds of 40 launches, the MEDIAN round reported (all three printed): a single 50-launch average after three warm-up launches sat on the
    # clock ramp of a chip that had just idled through the operand set-up -- 221 / 235 us in two processes against 191-199 us in tools/attn_split3_time.py
    n, rounds = 40, []
    for _ in range(n):
        ops.vit_attention(q3, out, B, S, Hh)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.vit_attention(q3, out, B, S, Hh)
        e1.record()
        e1.synchronize()
        rounds.append(e0.elapsed_time(e1) / n)
    ms = sorted(rounds)[1]
    flops = 4.0 * B * Hh * S * S * 64
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS["bf16"] / 6.0
    return {"kernel": "vit_attention_split3_pipe_kernel (32 x 32 x 16 bf16 MFMAs x 6 per float32 product, QK / softmax / PV of three consecutive 32-key blocks overlapped "
                      f"in every wave; LDS-DMA K / V rings, transposing V reads) at {B} tiles x {Hh} heads x {S} tokens (24 such launches per tile batch)",
            "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "us_per_launch": round(ms * 1e3, 1),
            "rounds_us": [round(r * 1e3, 1) for r in rounds], "flops_per_launch": flops}


def roofline(dtype, dev, gemm_only=False):
    """Dominant kernel of the pass, timed live through the entry points the engine uses (HIP events on the launch stream).

    bf16: the 3x3 halo kernel on the fusion U-Net's 544->544 conv @ [8,392,518]; algorithmic FLOPs = 2 * 8*392*518 * 9*544 * 544 per
    launch.
    fp32: the layer runs as Winograd F(4x4, 3x3); whichever form the engine's dispatch picks for it (hip_ops._fused_wanted) is timed:
    the batched transform-domain GEMM in split precision (default), the fused kernel (PF_WINO_SPLIT3=0), or the batched f32 GEMM
    (PF_WINO_FUSED=0 too): FLOPs = 36 * 2 * T * 544 * 544 with T = 8 * ceil(392/4) * ceil(518/4) tiles -- the layer's OWN useful
    multiply-adds in the transform domain, priced against the f32 MFMA peak.  `layer` adds the whole layer: its time and the
    direct-convolution FLOPs it replaces per second (which may exceed the MFMA peak: Winograd multiplies 36 / 144 as often).
    PF_WINOGRAD=0: the direct f32 kernel on the 3x3 layer, as in bf16.
    `traffic` (HBM bytes per launch from the rocprofv3 PMC passes, which cannot run inside bench.py) is reported only when
    profiles/r5_pmc_dominant_<dtype>.json was measured on EXACTLY this kernel source (sha of igemm.hip + wino_fused.hip + gemm_split3.hip + winograd.hip + pf_common.h); otherwise null.
    NOTE on its meaning: FETCH_SIZE / WRITE_SIZE count the L2's fabric-side requests; reads served by the 256 MiB Infinity Cache are included, so
    for the fused Winograd kernel (whose 43 MB filter set and 10 MB halo groups are re-streamed through L2 by design) it is an UPPER bound on HBM bytes."""
    from patchfusion_amd import hip_ops
    from patchfusion_amd import packing as pk
    from patchfusion_amd.hip_ops import ops
    hip_ops.refresh_env()             # (the PF_* switches are cached per engine build; this function runs outside one)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    B, H, W, C = 8, 392, 518, 544
    w = torch.randn(C, C, 3, 3) / (9 * C) ** 0.5
    pw = pk.pack_conv(w, torch.zeros(C), dtype=tdt).to(dev)
    peak = PEAK_TFLOPS[dtype]
    traffic = None
    for rnd in ("r6", "r5", "r4", "r3"):        # the newest PMC summary measured on EXACTLY this kernel source
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_dominant_{dtype}.json")) as f:
                j = json.load(f)
            if j.get("kernel_source_sha") == kernel_source_sha() and j.get("winograd_m", 0) == pw.wino_m:
                traffic = j["derived"]["traffic_bytes"]
                break
        except Exception:
            pass
    direct_flops = 2.0 * B * H * W * 9 * C * C
    if (pw.wino_u3 is not None and pk.winograd_applies(pw, B * H * W, 1, 1, "relu") and hip_ops._split3_three_step(pw) and
            not hip_ops._fused_wanted(B, H, W, pw)):
        # round 3 (late): the layer runs as input transform (three bf16 planes) -> ONE batched split-precision GEMM launch over the 36 transform
        # points -> output transform.  Dominant launch = that GEMM (round 4: gemm_split3_persist192_kernel, v_mfma_f32_16x16x32_bf16 x 6 per useful product):
        # priced as its USEFUL float32 multiply-adds 36 * 2 * T * C * C against the f32 MFMA peak -- the roofline of the arithmetic the layer
        # asks for -- with the executed bf16 rate against the bf16 peak beside it.
        # (round 5: the engine walks a layer whose V + M arenas would exceed PF_WS_CAP_GB in windows of its Winograd tiles, hip_ops.wino3_window -- the launch
        # timed here is the launch the pass issues: `window` tiles per launch, ns launches per layer call)
        T, ns = hip_ops.wino3_window(B, H, W, pw)[:2]
        Tall = B * -(-H // 4) * -(-W // 4)
        T = min(T, Tall)
        V3 = torch.randn(3, 36, C // 32, T, 32, device=dev).to(torch.bfloat16)       # chunk-major planes, as the input transform writes them
        Mw = torch.empty(36 * T * C, device=dev)
        iters = 5 * ns                    # (the same ~50 ms of launches whatever the sub-batch: a 13 ms window of short launches reads ~10 % slow -- clock ramp)
        ms = ops.gemm_planes_split3_timed(V3, pw.wino_u3, Mw.view(36, T, C), T, C, C, iters)
        flops = 36 * 2.0 * T * C * C
        ach = flops / (ms * 1e-3) / 1e12
        del V3, Mw
        ms_layer = None
        if not gemm_only:                 # (--roofline-only = PMC passes: the GEMM launches alone)
            x = torch.randn(B, H, W, C, device=dev)
            y = torch.empty(B, H, W, C, device=dev)
            ms_layer = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
        traffic = None
        for rnd in ("r6", "r5"):
            try:
                with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_dominant_fp32.json")) as f:
                    j = json.load(f)
                if j.get("kernel_source_sha") == kernel_source_sha() and abs(float(j.get("ws_cap_gb", -1)) - float(os.environ.get("PF_WS_CAP_GB", "2.5"))) < 1e-9:
                    traffic = j["derived"]["traffic_bytes"]
                    break
            except Exception:
                pass
        # the pipe this launch executes on is the bf16 MFMA, six instructions per useful float32 product: its ceiling for float32-grade work
        # is 2500 / 6 = 416.7 TF/s, and THAT is `peak` (round-3 review: pricing it against the 157.3 TF/s f32 MFMA peak, a pipe the kernel never
        # touches, printed 0.96 for a launch whose matrix pipe was 48 % busy)
        peak = PEAK_TFLOPS["bf16"] / 6.0
        return {"bound": "mfma",
                "kernel": f"gemm_split3_persist192_kernel (persistent 192 x 192 tiles, 6 x v_mfma_f32_16x16x32_bf16 per float32 product, f32 accumulation) as the batched transform-domain GEMM of the "
                          f"largest layer: 36 planes x [{T} x {C}].[{C} x {C}] = a window of {T} of the {Tall} Winograd tiles of 3x3 {C}->{C} @ {B}x{H}x{W} (GuidedFusion up-conv under F(4x4,3x3); "
                          f"the layer call issues {ns} such launches: workspace cap PF_WS_CAP_GB)",
                "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                # the dense bf16 MFMA rate this chip SUSTAINS on random operands, measured with a register-resident stream (tools/mfma_ceiling.hip,
                # profiles/r5_mfma_ceiling.md: 2020 TF/s at 2.05 GHz with the h / m / l operand pattern; zeros 2467 at 2.39 GHz) -- the practical ceiling beside the
                # nominal one.  `frac` stays against the nominal peak.
                "practical_peak": round(PRACTICAL_BF16_TFLOPS / 6.0, 1), "frac_of_practical_peak": round(ach / (PRACTICAL_BF16_TFLOPS / 6.0), 4),
                "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "launches_per_layer_call": ns, "calls": iters + 1, "traffic": traffic, "winograd_m": 4,
                "note": "achieved = USEFUL float32 FLOPs of the launch per second; peak = the dense bf16 MFMA peak / 6 (six bf16 MFMAs per float32-grade "
                        "product), so frac = executed bf16 FLOPs (padding columns not counted) over the 2.5 PF/s bf16 peak",
                "vs_f32_mfma_peak": round(ach / PEAK_TFLOPS["fp32"], 4),
                "executed_bf16": {"tflops": round(6 * ach, 1), "peak": PEAK_TFLOPS["bf16"], "frac": round(6 * ach / PEAK_TFLOPS["bf16"], 4)},
                "layer": None if ms_layer is None else {
                    "what": f"whole layer = input transform (split planes) + batched GEMM + output transform/epilogue, 3x3 {C}->{C} @ {B}x{H}x{W} ({ns} tile windows)",
                    # (GEMM share of the layer = the timed window scaled by tiles, Tall / T windows' worth: the last window is ragged -- round-5 advisor)
                    "ms": round(ms_layer, 4), "transforms_ms": round(ms_layer - ms * Tall / T, 4), "direct_conv_flops": direct_flops,
                    "direct_conv_tflops_equivalent": round(direct_flops / (ms_layer * 1e-3) / 1e12, 2),
                    # the WHOLE layer against the same ceiling: its transform-domain multiply-adds (all Tall tiles) over its whole time, transforms included
                    "frac": round(36 * 2.0 * Tall * C * C / (ms_layer * 1e-3) / 1e12 / peak, 4)},
                "layer_frac": None if ms_layer is None else round(36 * 2.0 * Tall * C * C / (ms_layer * 1e-3) / 1e12 / peak, 4)}
    if pw.wino_up is not None and pk.winograd_applies(pw, B * H * W, 1, 1, "relu") and hip_ops._fused_wanted(B, H, W, pw):
        # round 3: the layer is ONE kernel (csrc/wino_fused.hip): its own multiply-adds = 36 transform points x 2 x T x C x C with
        # T = B * ceil(H/4) * ceil(W/4) output tiles (the padding tiles of the 4x8 super-tiles are not counted), priced against the
        # f32 MFMA peak; the kernel also does both transforms and the epilogue, so kernel == layer
        T = B * -(-H // 4) * -(-W // 4)
        x = torch.randn(B, H, W, C, device=dev)
        y = torch.empty(B, H, W, C, device=dev)
        ms = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
        flops = 36 * 2.0 * T * C * C
        ach = flops / (ms * 1e-3) / 1e12
        try:
            with open(os.path.join(ROOT, "profiles", "r3_pmc_dominant_fp32.json")) as f:
                j = json.load(f)
            traffic = j["derived"]["traffic_bytes"] if j.get("kernel_source_sha") == kernel_source_sha() else None
        except Exception:
            traffic = None
        return {"bound": "mfma",
                "kernel": f"wino_fused_kernel (v_mfma_f32_16x16x4_f32): fused Winograd F(4x4,3x3) of the largest layer, 3x3 {C}->{C} @ {B}x{H}x{W} "
                          f"(GuidedFusion up-conv): input transform + 36 x [{T} x {C}].[{C} x {C}] + output transform + epilogue in one launch",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic, "winograd_m": 4,
                "layer": {"what": "kernel == whole layer (no separate transform passes)", "ms": round(ms, 4),
                          "direct_conv_flops": direct_flops, "direct_conv_tflops_equivalent": round(direct_flops / (ms * 1e-3) / 1e12, 2)}}
    if pw.wino_u is not None and pk.winograd_applies(pw, B * H * W, 1, 1, "relu"):
        m = pw.wino_m
        a2 = (m + 2) ** 2
        T = B * -(-H // m) * -(-W // m)
        V = torch.randn(a2 * T * C, device=dev)
        Mw = torch.empty(a2 * T * C, device=dev)
        ms = ops.gemm_planes_timed(V, pw.wino_u, Mw, a2, T, C, C, 5)
        flops = a2 * 2.0 * T * C * C
        ach = flops / (ms * 1e-3) / 1e12
        out = {"bound": "mfma",
               "kernel": f"conv_igemm_kernel<float> (v_mfma_f32_16x16x4_f32) as the batched Winograd-domain GEMM of the largest layer: "
                         f"{a2} planes x [{T} x {C}].[{C} x {C}] = 3x3 {C}->{C} @ {B}x{H}x{W} (GuidedFusion up-conv) under F({m}x{m},3x3)",
               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
               "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "traffic": traffic, "winograd_m": m}
        del V, Mw
        if not gemm_only:
            x = torch.randn(B, H, W, C, device=dev)
            y = torch.empty(B, H, W, C, device=dev)
            ms_layer = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
            out["layer"] = {"what": f"whole layer = input transform + batched GEMM + output transform/epilogue, 3x3 {C}->{C} @ {B}x{H}x{W}",
                            "ms": round(ms_layer, 4), "transforms_ms": round(ms_layer - ms, 4), "direct_conv_flops": direct_flops,
                            "direct_conv_tflops_equivalent": round(direct_flops / (ms_layer * 1e-3) / 1e12, 2)}
        return out
    x = torch.randn(B, H, W, C, device=dev).to(tdt)
    y = torch.empty(B, H, W, C, device=dev, dtype=tdt)
    ms = ops.conv(x, pw, y, pad=1, act="relu", _timed=5)
    ach = direct_flops / (ms * 1e-3) / 1e12
    kern = {"bf16": "conv3x3_halo_kernel (bf16, v_mfma_f32_32x32x16_bf16)", "fp32": "conv3x3 f32 kernel (v_mfma_f32_16x16x4_f32)"}[dtype]
    return {"bound": "mfma", "kernel": f"{kern}: 3x3 544->544 @ 8x392x518, the GuidedFusion up-conv = largest op",
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "ms_per_launch": round(ms, 4), "flops_per_launch": direct_flops, "traffic": traffic}


def cpu_baseline(cfg, sd, img):
    """CPU path on this box's host cores, bounded sample (threads capped at 64: with all 256 hardware threads of the GPU box
    torch's CPU kernels oversubscribe -- measured 108 s instead of ~5 s for one branch forward).
    kind 'reference': the reference's own `PatchFusion` (imported through oracle/ref_shim.py) on a 1x1-tile 4K image = ONE
        patch through coarse branch + fine branch + fusion (the G2L stacks inside the fusion call, reference schedule);
        only possible where /root/reference exists (not on the GPU box).
    kind 'port': the oracle restatement, ONE tile of the same workload = fine branch + fusion_forward (4029.9 GFLOP, G2L
        hoisted like the engine does); the coarse pass + G2L it needs are computed untimed."""
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    from oracle import ref_shim
    if ref_shim.reference_available():
        try:
            return _cpu_baseline_reference(cfg, sd, img, cores)
        except Exception as e:      # a broken reference import must not lose the whole bench line
            log(f"cpu baseline: reference path failed ({type(e).__name__}: {e}); falling back to the port")
    from oracle import pf_oracle
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
    out = {"value": round(1.0 / dt, 5), "unit": "patches/s", "cores": cores, "kind": "port",
           "sample": f"1 tile (fine branch + fusion, 4029.9 GFLOP, G2L hoisted) of the same DA-vitl 392x518 workload: "
                     f"{dt:.1f} s on {cores} threads"}
    try:      # the reference's OWN code cannot run on the GPU box (/root/reference is absent there): the figure recorded in the build container
        with open(os.path.join(ROOT, "profiles", "r3_cpu_reference.json")) as f:
            out["reference_recorded"] = json.load(f)
    except Exception:
        pass
    return out


def _cpu_baseline_reference(cfg, sd, img, cores):
    from oracle import ref_shim
    ref_shim.install_stubs()
    c1 = dict(cfg)
    c1["patch_split_num"] = (1, 1)
    with ref_shim.in_reference_cwd():
        PF = ref_shim.import_reference()
        m = PF(c1).eval()
        m.load_state_dict(sd, strict=True)
        with torch.no_grad():
            lr = m.resizer(img)
            t0 = time.perf_counter()
            m(mode="infer", image_lr=lr, image_hr=img, cai_mode="m1", process_num=1)
            dt = time.perf_counter() - t0
    log(f"cpu baseline: reference PatchFusion, one 1x1-tile image {dt:.1f}s on {cores} threads")
    return {"value": round(1.0 / dt, 5), "unit": "patches/s", "cores": cores, "kind": "reference",
            "sample": f"the reference's own PatchFusion.forward(mode='infer') on one 2160x3840 image with patch_split_num=(1,1): 1 patch "
                      f"= coarse branch + fine branch + fusion incl. G2L (5373 GFLOP): {dt:.1f} s on {cores} threads"}


if __name__ == "__main__":
    main()
