"""Patch-level sharding of ONE image over the GPUs of a node.

After the coarse branch (+G2L) -- replicated on every rank, no communication -- every tile is
independent, so ranks take contiguous chunks of the tile list and the only collective is a gather of
the per-tile depths ([P_r, h, w] float32, 0.81 MB per tile at 392x518).  With RCCL over xGMI every
peer owns a direct link, so one all_gather of <= 6.5 MB per rank is far below the per-link bound; no
ring all-reduce is involved.  Every rank then stitches the full map in the reference's tile order
(deterministic: identical result on all ranks).  backend 'nccl' == RCCL on ROCm; CPU tests use gloo.
"""
import torch
import torch.distributed as dist

from .tiling import shard_range


def all_gather_shards(preds, n, world):
    """preds [n,h,w]: this rank has filled rows shard_range(n, rank, world); returns the full tensor."""
    rank = dist.get_rank()
    q = (n + world - 1) // world                     # padded equal-size chunks for all_gather_into_tensor
    lo, hi = shard_range(n, rank, world)
    send = torch.zeros((q,) + tuple(preds.shape[1:]), dtype=preds.dtype, device=preds.device)
    send[:hi - lo] = preds[lo:hi]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)                      # RCCL (backend 'nccl') on GPUs, gloo in CPU tests
    out = torch.empty_like(preds)
    for r in range(world):
        l, h = shard_range(n, r, world)
        out[l:h] = recv[r][:h - l]
    return out
