"""N>1 path on CPU: two gloo ranks shard the tiles of one image, all-gather the per-tile depths and
both stitch the identical map that a single process produces (engine wiring via tests/fake_ops.py)."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from patchfusion_amd import tiling
from patchfusion_amd.config import make_config
from patchfusion_amd.spec import patchfusion_spec, synthetic_state_dict

TINY = ("vits", (112, 154), (448, 616), (2, 2))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from patchfusion_amd.model import PatchFusion
    from tests.fake_ops import ops as fake_ops
    cfg = make_config(*TINY)
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops).eval()
    m.load_state_dict(sd, strict=True)
    img = torch.rand(1, 3, *TINY[2], generator=torch.Generator().manual_seed(1234))
    random.seed(5621)
    # tools/test.py:221 wraps the model in DistributedDataParallel before Tester.run calls it
    ddp = torch.nn.parallel.DistributedDataParallel(m)
    with torch.no_grad():
        d, _ = ddp(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode=mode, process_num=2)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), d.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["m2"])
def test_two_rank_patch_sharding_matches_golden(tmp_path, golden_dir, mode):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)                                   # every rank stitches the same map
    ref = np.load(os.path.join(golden_dir, "tiny_vits.npz"))[f"depth_{mode}"]
    assert np.abs(a[0, 0] - ref).max() < 2e-5


def test_shard_range_partitions_exactly():
    for n in (1, 13, 16, 49, 64, 177):
        for w in (1, 2, 4, 8):
            r = [tiling.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_tile_schedule_counts_match_reference_docs():
    """docs/user_infer.md:22-37: 4x4 split -> m1 = 16 tiles, m2 = 49, r128 = 49 + 128."""
    tc = tiling.prepare_tile_cfg((392, 518), (2160, 3840), (4, 4))
    assert len(tiling.tile_schedule(tc, (392, 518), "m1", 4)) == 16
    assert len(tiling.tile_schedule(tc, (392, 518), "m2", 4)) == 49
    random.seed(0)
    t = tiling.tile_schedule(tc, (392, 518), "r128", 4)
    assert len(t) == 177 and sum(x["phase"] == "random" for x in t) == 128
    # random tiles of one call share ONE w_start (baseline_pretrain.py:155-156)
    rnd = [x["box"][0] for x in t if x["phase"] == "random"]
    assert all(len(set(rnd[i:i + 4])) == 1 for i in range(0, 128, 4))
    with pytest.raises(ValueError):
        tiling.tile_schedule(tc, (392, 518), "zz", 4)
    assert tc["patch_raw_shape"] == (540, 960) and tc["patch_reensemble_shape"] == (1568, 2072)


def test_blend_mask_matches_oracle():
    from oracle.pf_oracle import generatemask
    for size in ((112, 154), (224, 308)):
        assert np.abs(tiling.gaussian_blend_mask(size) - generatemask(size)).max() < 1e-6
