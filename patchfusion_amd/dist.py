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


_GATHER_BUF = {}


def all_gather_shards(preds, n, world):
    """preds [n,h,w]: this rank has filled rows shard_range(n, rank, world); returns the full tensor (every rank: the same bits).

    ONE collective on a preallocated buffer.  Equal shards (n % world == 0: BASELINE configs[3] -- 64 tiles over 8 GPUs -- and every
    `bench.py --gpus N` split): `all_gather_into_tensor` writes rank r's chunk at rows [r q, (r+1) q) of the result, which IS the tile order --
    no staging copy on either side (round 3 made a padded send copy, a Python list of `world` receive tensors and `world` scatter copies).
    Ragged shards: the chunks are padded to the largest one in a cached [world, q, h, w] buffer and compacted with one indexed copy."""
    rank = dist.get_rank()
    lo, hi = shard_range(n, rank, world)
    q, r = divmod(n, world)
    if r == 0:
        out = torch.empty_like(preds)
        dist.all_gather_into_tensor(out, preds[lo:hi].contiguous())       # RCCL (backend 'nccl') on GPUs, gloo in CPU tests
        return out
    qp = q + 1
    # n is part of the key: the compaction index `rows` depends on it (two ragged tile counts with the same qp -- world 4: n = 9 then 10 -- would
    # otherwise reuse a stale index and return the wrong number of rows)
    key = (world, n, tuple(preds.shape[1:]), preds.dtype, preds.device)
    buf = _GATHER_BUF.get(key)
    if buf is None:
        if len(_GATHER_BUF) > 4:
            _GATHER_BUF.clear()
        rows = torch.cat([torch.arange(l, h) + (rr * qp - l) for rr in range(world) for l, h in [shard_range(n, rr, world)]])
        buf = _GATHER_BUF[key] = (torch.zeros((world * qp,) + tuple(preds.shape[1:]), dtype=preds.dtype, device=preds.device),
                                  torch.zeros((qp,) + tuple(preds.shape[1:]), dtype=preds.dtype, device=preds.device), rows.to(preds.device))
    recv, send, rows = buf
    send[:hi - lo] = preds[lo:hi]
    dist.all_gather_into_tensor(recv, send)
    return recv.index_select(0, rows)
