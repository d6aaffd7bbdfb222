"""Tile geometry, blend masks and the tile schedule of PatchFusion inference (host-side control flow).

Restates (never imports) the reference's
  estimator/models/baseline_pretrain.py:91-119   prepare_tile_cfg
  estimator/models/baseline_pretrain.py:222-331  regular_tile   (grid enumeration / paste positions)
  estimator/models/baseline_pretrain.py:143-218  random_tile    (RNG draw order: process_num h, one w)
  estimator/models/utils.py:38-47                generatemask
The arithmetic on pixels (crop+resize, stitch) runs in HIP kernels; this module only produces the
small integer/float tables they consume.
"""
import functools
import random

import numpy as np


def prepare_tile_cfg(process_shape, image_raw_shape, patch_split_num):
    assert image_raw_shape[0] % (2 * patch_split_num[0]) == 0, \
        'image height should be divisible by 2 * patch_split_num[0]'
    assert image_raw_shape[1] % (2 * patch_split_num[1]) == 0, \
        'image width should be divisible by 2 * patch_split_num[1]'
    patch_raw_shape = (image_raw_shape[0] // patch_split_num[0], image_raw_shape[1] // patch_split_num[1])
    return {
        'patch_split_num': tuple(patch_split_num),
        'patch_reensemble_shape': (process_shape[0] * patch_split_num[0], process_shape[1] * patch_split_num[1]),
        'patch_raw_shape': patch_raw_shape,
        'image_raw_shape': tuple(image_raw_shape),
        'raw_h_split_point': [int(patch_raw_shape[0] * i) for i in range(patch_split_num[0])],
        'raw_w_split_point': [int(patch_raw_shape[1] * i) for i in range(patch_split_num[1])],
    }


@functools.lru_cache(maxsize=16)
def gaussian_blend_mask(size):
    """Blend mask of one tile: ones in the central 80 %, Gaussian-blurred (sigma = int(H/16),
    k = 2*ceil(2*sigma)+1, BORDER_REFLECT_101 like cv2.GaussianBlur) and min-max normalised.
    Constant per shape -> computed once on the host and cached (the reference recomputes it on the
    CPU in every forward, patchfusion.py:415)."""
    h, w = int(size[0]), int(size[1])
    mask = np.zeros((h, w), dtype=np.float32)
    sigma = int(h / 16)
    k = int(2 * np.ceil(2 * int(h / 16)) + 1)
    mask[int(0.1 * h):h - int(0.1 * h), int(0.1 * w):w - int(0.1 * w)] = 1
    xs = np.arange(k, dtype=np.float64) - (k - 1) * 0.5
    kern = np.exp(-(xs * xs) / (2.0 * sigma * sigma))
    kern /= kern.sum()
    kern = kern.astype(np.float32).astype(np.float64)
    r = k // 2
    pad = np.pad(mask.astype(np.float64), ((0, 0), (r, r)), mode='reflect')
    tmp = np.zeros((h, w), dtype=np.float64)
    for i in range(k):
        tmp += kern[i] * pad[:, i:i + w]
    tmp = tmp.astype(np.float32).astype(np.float64)
    pad = np.pad(tmp, ((r, r), (0, 0)), mode='reflect')
    out = np.zeros((h, w), dtype=np.float64)
    for i in range(k):
        out += kern[i] * pad[i:i + h, :]
    out = out.astype(np.float32)
    out = (out - out.min()) / (out.max() - out.min())
    return out.astype(np.float32)


def regular_grid(tile_cfg, process_shape, offset, offset_process):
    """Boxes (x0,y0,x1,y1) in raw pixels and paste positions (y,x) in re-ensemble pixels of one
    regular grid, row-major like the reference's nested loops."""
    height, width = tile_cfg['patch_raw_shape']
    H, W = tile_cfg['image_raw_shape']
    assert offset[0] >= 0 and offset[1] >= 0 and offset_process[0] >= 0 and offset_process[1] >= 0
    hs = [height * i + offset[0] for i in range((H - offset[0]) // height)]
    ws = [width * i + offset[1] for i in range((W - offset[1]) // width)]
    RH, RW = tile_cfg['patch_reensemble_shape']
    hp = [process_shape[0] * i + offset_process[0] for i in range((RH - offset_process[0]) // process_shape[0])]
    wp = [process_shape[1] * i + offset_process[1] for i in range((RW - offset_process[1]) // process_shape[1])]
    boxes = [(w, h, w + width, h + height) for h in hs for w in ws]
    paste = [(h, w) for h in hp for w in wp]
    assert len(boxes) == len(paste)
    return boxes, paste


def tile_schedule(tile_cfg, process_shape, cai_mode, process_num):
    """All tiles of one image in the reference's processing order.
    Returns a list of dicts {phase: 'init'|'regular'|'random', box, paste}.  Random tiles draw from
    python's `random` exactly like random_tile does (process_num h_starts then ONE shared w_start per
    call, int(N)//process_num calls) so a seeded run reproduces the reference's tile positions."""
    ps = process_shape
    tiles = []
    boxes, paste = regular_grid(tile_cfg, ps, (0, 0), (0, 0))
    tiles += [dict(phase='init', box=b, paste=p) for b, p in zip(boxes, paste)]
    if cai_mode == 'm2' or cai_mode[0] == 'r':
        hr, wr = tile_cfg['patch_raw_shape']
        for off, offp in (((0, wr // 2), (0, ps[1] // 2)), ((hr // 2, 0), (ps[0] // 2, 0)),
                          ((hr // 2, wr // 2), (ps[0] // 2, ps[1] // 2))):
            boxes, paste = regular_grid(tile_cfg, ps, off, offp)
            tiles += [dict(phase='regular', box=b, paste=p) for b, p in zip(boxes, paste)]
    if cai_mode[0] == 'r':
        height, width = tile_cfg['patch_raw_shape']
        H, W = tile_cfg['image_raw_shape']
        for _ in range(int(cai_mode[1:]) // process_num):
            hs = [random.randint(0, H - height - 1) for _ in range(process_num)]
            ws = [random.randint(0, W - width - 1)]
            for h in hs:
                for w in ws:
                    tiles.append(dict(phase='random', box=(w, h, w + width, h + height), paste=(h, w)))
    elif cai_mode not in ('m1', 'm2'):
        raise ValueError(f"unknown cai_mode {cai_mode!r} (expected m1, m2 or r<int>)")
    return tiles


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n tiles for `rank` (earlier ranks take the remainder)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)
