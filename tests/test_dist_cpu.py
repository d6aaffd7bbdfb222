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


def _model(shard):
    from patchfusion_amd.model import PatchFusion
    from tests.fake_ops import ops as fake_ops
    cfg = make_config(*TINY)
    if shard == "config":                       # the mmengine-config route: MODELS.build(cfg.model) cannot pass kwargs
        cfg["shard_patches"] = True
    sd = synthetic_state_dict(patchfusion_spec(cfg), 0)
    m = PatchFusion(cfg, compute_dtype="fp32", ops=fake_ops, **({"shard_patches": True} if shard == "kwarg" else {})).eval()
    m.load_state_dict(sd, strict=True)
    return m


def _image(seed):
    return torch.rand(1, 3, *TINY[2], generator=torch.Generator().manual_seed(seed))


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _worker_shard(rank, world, port, mode, shard, out_dir):
    """single-image tile sharding (explicit opt-in): both ranks hold the SAME image."""
    _init(rank, world, port)
    m = _model(shard)
    img = _image(1234)
    # r-mode: the ranks deliberately start from DIFFERENT python-random states; rank 0's schedule must win
    random.seed(5621 if rank == 0 else 777 + rank)
    # tools/test.py:221 wraps the model in DistributedDataParallel before Tester.run calls it
    ddp = torch.nn.parallel.DistributedDataParallel(m)
    with torch.no_grad():
        d, _ = ddp(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode=mode, process_num=2)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), d.numpy())
    dist.destroy_process_group()


def _worker_shard_8x8(rank, world, port, out_dir):
    """BASELINE configs[3] in miniature: an 8x8 grid (64 tiles) of ONE image sharded over the ranks, 32 tiles per rank."""
    _init(rank, world, port)
    m = _model("kwarg")
    img = torch.rand(1, 3, 448, 640, generator=torch.Generator().manual_seed(77))
    tc = {"image_raw_shape": (448, 640), "patch_split_num": (8, 8)}
    with torch.no_grad():
        d, _ = m(mode="infer", image_lr=m.resizer(img), image_hr=img, tile_cfg=tc, cai_mode="m1", process_num=8)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), d.numpy())
    dist.destroy_process_group()


def _worker_image_dp(rank, world, port, n_images, out_dir):
    """the reference's own multi-GPU evaluation (tools/test.py:218-239, tester.py:46-64): DDP + DistributedSampler,
    every rank a DIFFERENT image, default-constructed model -> no tile sharding, no collective, correct maps.  With an odd
    image count the sampler pads by repeating image 0; nothing may hang."""
    from torch.utils.data import DataLoader, Dataset
    from torch.utils.data.distributed import DistributedSampler

    class Images(Dataset):
        def __len__(self):
            return n_images

        def __getitem__(self, i):
            return dict(idx=i, image_hr=_image(100 + i)[0])

    _init(rank, world, port)
    m = _model(None)
    assert m.shard_patches is False
    ddp = torch.nn.parallel.DistributedDataParallel(m)
    ds = Images()
    loader = DataLoader(ds, batch_size=1, sampler=DistributedSampler(ds, shuffle=False))
    for batch in loader:
        hr = batch["image_hr"]
        with torch.no_grad():
            d, _ = ddp(mode="infer", image_lr=m.resizer(hr), image_hr=hr, cai_mode="m1", process_num=2)
        np.save(os.path.join(out_dir, f"img{int(batch['idx'])}_rank{rank}.npy"), d.numpy())
    dist.destroy_process_group()


def _worker_mismatch(rank, world, port, out_dir):
    """opt-in sharding but the ranks hold different images: every rank must raise (nobody hangs in the gather)."""
    _init(rank, world, port)
    m = _model("kwarg")
    img = _image(1234 + rank)
    try:
        with torch.no_grad():
            m(mode="infer", image_lr=m.resizer(img), image_hr=img, cai_mode="m1", process_num=2)
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(msg)
    dist.destroy_process_group()


def _worker_ragged_gather(rank, world, port, out_dir):
    """two ragged tile counts with the SAME padded shard size in one process (world 3: n = 4 then 5, both pad to 2 rows per rank;
    then the equal-shard n = 6 and n = 4 again): the compaction index must follow n, not the padded size."""
    _init(rank, world, port)
    from patchfusion_amd.dist import all_gather_shards
    ok = True
    for n in (4, 5, 6, 4, 7, 8):
        full = torch.arange(n * 6, dtype=torch.float32).view(n, 2, 3) + 1
        lo, hi = tiling.shard_range(n, rank, world)
        mine = torch.zeros_like(full)
        mine[lo:hi] = full[lo:hi]
        got = all_gather_shards(mine, n, world)
        ok = ok and got.shape == full.shape and torch.equal(got, full)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_ragged_gather_two_tile_counts_with_equal_padded_shard_three_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker_ragged_gather, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(3)] == ["ok"] * 3


@pytest.mark.parametrize("mode,shard", [("m2", "kwarg"), ("m1", "config"), ("r4", "kwarg")])
def test_two_rank_patch_sharding_matches_golden(tmp_path, golden_dir, mode, shard):
    port = _free_port()
    mp.spawn(_worker_shard, args=(2, port, mode, shard, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)                                   # every rank stitches the same map
    ref = np.load(os.path.join(golden_dir, "tiny_vits.npz"))[f"depth_{mode}"]   # r4 golden: random.seed(5621) = rank 0's state
    assert a[0, 0].shape == ref.shape and np.abs(a[0, 0] - ref).max() < 2e-5


def test_8x8_grid_sharded_two_ways_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker_shard_8x8, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b) and a.shape == (1, 1, 8 * 112, 8 * 154)
    single = _model(None)
    img = torch.rand(1, 3, 448, 640, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        ref, _ = single(mode="infer", image_lr=single.resizer(img), image_hr=img,
                        tile_cfg={"image_raw_shape": (448, 640), "patch_split_num": (8, 8)}, cai_mode="m1", process_num=8)
    assert np.abs(a - ref.numpy()).max() < 2e-5


@pytest.mark.parametrize("n_images", [2, 3])
def test_image_level_data_parallel_like_tools_test_py(tmp_path, n_images):
    port = _free_port()
    mp.spawn(_worker_image_dp, args=(2, port, n_images, str(tmp_path)), nprocs=2, join=True)
    single = _model(None)
    for i in range(n_images):
        files = sorted(tmp_path.glob(f"img{i}_rank*.npy"))
        assert files, f"image {i} was never evaluated"
        img = _image(100 + i)
        with torch.no_grad():
            ref, _ = single(mode="infer", image_lr=single.resizer(img), image_hr=img, cai_mode="m1", process_num=2)
        for f in files:
            # each rank's map == the single-process map of ITS image (thread count differs -> summation order, not bits)
            assert np.abs(np.load(f) - ref.numpy()).max() < 2e-5, f.name


def test_sharding_with_different_images_raises_on_every_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker_mismatch, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "DIFFERENT images" in (tmp_path / f"rank{r}.txt").read_text()


def test_sharding_is_off_by_default_and_env_opt_in(monkeypatch):
    assert _model(None).shard_patches is False
    monkeypatch.setenv("PF_SHARD_PATCHES", "1")
    assert _model(None).shard_patches is True


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
