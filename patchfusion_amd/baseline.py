"""``BaselinePretrain`` -- the reference's single-branch models:
``target='coarse'``: the coarse ZoeDepth/Depth-Anything branch on the down-sampled image;
``target='fine'``  : the fine branch tiled over the 4K image and stitched, WITHOUT the fusion network
(estimator/models/baseline_pretrain.py:44-89 constructor, :121-141 load/save, :333-420 forward).
Same HIP engine (BranchNet + stitcher); checkpoints hold the branch keys without prefix
(``self.coarse_branch.load_state_dict(dict, strict=True)``, baseline_pretrain.py:121-127).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import tiling
from .config import AttrDict
from .model import PatchFusion, Resizer, _DTYPES, _build_param_tree
from .spec import branch_spec


class BaselinePretrain(PatchFusion):
    def __init__(self, coarse_branch, fine_branch, sigloss=None, min_depth=1e-3, max_depth=80, image_raw_shape=(2160, 3840),
                 patch_process_shape=(384, 512), patch_split_num=(4, 4), target='coarse', coarse_branch_zoe=None,
                 compute_dtype=None, ops=None, core_provider=None):
        nn.Module.__init__(self)
        if target not in ('coarse', 'fine'):
            raise NotImplementedError(target)
        self.target = target
        self.patch_process_shape = tuple(patch_process_shape)
        self.tile_cfg = self.prepare_tile_cfg(image_raw_shape, patch_split_num)
        self.min_depth, self.max_depth = min_depth, max_depth
        self.sigloss_cfg = dict(sigloss) if isinstance(sigloss, dict) else None
        self.coarse_branch_cfg, self.fine_branch_cfg = AttrDict(dict(coarse_branch)), AttrDict(dict(fine_branch))
        self.branch_cfg = self.coarse_branch_cfg if target == 'coarse' else self.fine_branch_cfg
        # baseline_pretrain.py:67-86: 'DA-ZoeDepth' (Depth-Anything ViT core, resize multiple 14) or 'ZoeDepth' (MiDaS/BEiT core, an
        # un-vendored torch.hub repo: supplied as `core_provider`, see engine.ExternalCoreBranchNet; resize multiple 32)
        if self.branch_cfg.type == 'DA-ZoeDepth':
            if self.branch_cfg.midas_model_type not in ('vits', 'vitb', 'vitl'):
                raise NotImplementedError(self.branch_cfg.midas_model_type)
        elif self.branch_cfg.type != 'ZoeDepth':
            raise NotImplementedError
        self.core_provider = core_provider
        self.prefix = f"{target}_branch."
        self.resizer = Resizer(self.patch_process_shape[1], self.patch_process_shape[0], 32 if self.branch_cfg.type == 'ZoeDepth' else 14)
        self.spec = OrderedDict()
        branch_spec(self.spec, self.prefix, self.branch_cfg)
        _build_param_tree(self, self.spec)
        self.compute_dtype = _DTYPES[compute_dtype or "fp32"]
        self.shard_patches, self.overlap_coarse, self.overlap_batches, self.n_streams = False, False, False, 1
        self._ops, self._engine, self._coarse_state = ops, None, None
        self._side_stream, self._aux_streams = None, []

    def load_dict(self, dict):
        return self.load_state_dict({self.prefix + k: v for k, v in dict.items()}, strict=True)

    def get_save_dict(self):
        return OrderedDict((k[len(self.prefix):], v) for k, v in self.state_dict().items())

    def _ensure_engine(self):
        if self._engine is None:
            from .engine import BranchNet, ExternalCoreBranchNet
            sd = self.state_dict()
            dev = next(iter(sd.values())).device
            if self._ops is None and dev.type != "cuda":
                raise RuntimeError("BaselinePretrain (MI355X engine) needs the model on a GPU: call .cuda() first")
            if self.branch_cfg.type == 'ZoeDepth':
                net = ExternalCoreBranchNet(sd, self.prefix, self.branch_cfg, self.patch_process_shape, self.compute_dtype, dev, self.core_provider)
            else:
                net = BranchNet(sd, self.prefix, self.branch_cfg, self.patch_process_shape, self.compute_dtype, dev)
            self._engine = dict(branch=net)
            self._device, self._mask_cache, self._table_cache = dev, {}, {}
        return self._engine

    def infer_forward(self, imgs_crop):
        nets = self._ensure_engine()
        depth, _ = nets["branch"].forward(self.ops, imgs_crop.contiguous().float())
        return depth.unsqueeze(1)

    @torch.no_grad()
    def forward(self, mode, image_lr, image_hr, depth_gt=None, crop_depths=None, crops_image_hr=None, bboxs=None,
                tile_cfg=None, cai_mode='m1', process_num=4, **kwargs):
        nets = self._ensure_engine()
        ops, dev = self.ops, self._device
        if mode == 'train':
            # baseline_pretrain.py:347-363: the branch on the batch + SILogLoss.  FORWARD VALUE ONLY (no backward kernels), like
            # PatchFusion.train_forward.
            x, gt, key = (image_lr, depth_gt, 'coarse_loss') if self.target == 'coarse' else (crops_image_hr, crop_depths, 'fine_loss')
            depth, _ = nets["branch"].forward(ops, x.contiguous().float())
            depth = depth.unsqueeze(1)
            loss_dict = {key: self._sigloss(depth, gt, self.sigloss_cfg)}
            loss_dict['total_loss'] = loss_dict[key]
            return loss_dict, {'rgb': image_lr, 'depth_pred': depth, 'depth_gt': gt}
        if self.target == 'coarse':
            depth = self.infer_forward(image_lr)
            return depth, {'rgb': image_lr, 'depth_pred': depth, 'depth_gt': depth_gt}
        tile_cfg = self.tile_cfg if tile_cfg is None else self.prepare_tile_cfg(tile_cfg['image_raw_shape'], tile_cfg['patch_split_num'])
        assert image_hr.shape[0] == 1
        # baseline_pretrain.py:404-408: r<N> makes N random_tile calls (each process_num tiles)
        sched_mode = cai_mode if cai_mode[0] != 'r' else f"r{int(cai_mode[1:]) * process_num}"
        tiles = tiling.tile_schedule(tile_cfg, self.patch_process_shape, sched_mode, process_num)
        ph, pw = self.patch_process_shape
        preds = ops.empty((len(tiles), ph, pw), torch.float32, dev)
        img = image_hr[0].contiguous().float()
        bt, _ = self._tile_tables(tiles, tile_cfg)
        for s in range(0, len(tiles), process_num):
            e = min(s + process_num, len(tiles))
            crops = ops.empty((e - s, 3, ph, pw), torch.float32, dev)
            ops.crop_resize(img, bt[s:e], crops)
            depth, _ = nets["branch"].forward(ops, crops)
            ops.copy_plane(depth.unsqueeze(1), preds[s:e])
        avg = self._stitch(preds, tiles, tile_cfg, cai_mode)
        return avg.unsqueeze(0).unsqueeze(0), {}
