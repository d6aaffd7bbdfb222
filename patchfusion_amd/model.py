"""Drop-in boundary: ``PatchFusion`` with the reference's constructor, ``forward`` signature,
checkpoint keys and helper attributes (estimator/models/patchfusion.py:57-453,
estimator/models/baseline_pretrain.py:91-331), executing on the MI355X HIP engine.

What is mirrored:
  * ``PatchFusion(config)`` with the mmengine ``model.config`` mapping (ConfigDict or plain dict)
  * ``state_dict()`` keys/shapes identical to the reference (1030 tensors for vits) so reference
    ``.pth`` / HF checkpoints load by key; ``load_dict`` (strict=False) and ``get_save_dict``
  * ``forward(mode, image_lr, image_hr, depth_gt=None, crops_image_hr=None, crop_depths=None, bboxs=None,
    tile_cfg=None, cai_mode='m1', process_num=4)`` -> ``(depth [1,1,H',W'], {'rgb','depth_pred','depth_gt'})``
  * ``tile_cfg``, ``resizer``, ``patch_process_shape``, ``coarse_forward``, ``fine_forward``, ``infer_forward``
  * errors: AssertionError for divisibility / batch-1, NotImplementedError for unknown branch types,
    ValueError for bin_centers_type -- same exception types as the reference.
``mode='train'`` computes the reference's training-mode FORWARD (loss value, patchfusion.py:372-399) without autograd: the
engine has no backward kernels (north star is inference).

Multi-GPU: two axes, like the reference leaves them.
  * image-level data parallelism (tools/test.py:218-239: DDP + DistributedSampler, every rank holds a
    DIFFERENT image) is the DEFAULT -- the module does no communication at all;
  * single-image tile sharding (BASELINE configs[3]/[4]) is an explicit opt-in: ``shard_patches=True`` in the
    constructor, ``shard_patches=True`` in the model config, or ``PF_SHARD_PATCHES=1``.  Then the tiles of ONE
    image are sharded over ranks (contiguous chunks) and the per-patch depths are all-gathered over RCCL.
    Before sharding every forward verifies with one tiny all_reduce that all ranks really hold the same
    image (and, for ``r<N>``, rank 0's random tile schedule is broadcast) -- a mismatch raises on every rank
    instead of silently stitching tiles of different images.
"""
import os as _os
from collections import OrderedDict
from contextlib import nullcontext as _nullcontext

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import tiling
from .config import AttrDict
from .spec import patchfusion_spec

try:  # HF mixin keeps `from_pretrained` / `save_pretrained` available like the reference class
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
           torch.float32: torch.float32, torch.bfloat16: torch.bfloat16}


def _is_mmengine_configdict(config):
    """isinstance(config, mmengine.ConfigDict) without requiring mmengine (the reference's test, patchfusion.py:64)."""
    try:
        from mmengine import ConfigDict
        if isinstance(config, ConfigDict):
            return True
    except Exception:
        pass
    return any(c.__name__ == "ConfigDict" for c in type(config).__mro__)


class Resizer:
    """depth_anything/transform.py Resize(width, height, keep_aspect_ratio=False,
    ensure_multiple_of=m, resize_method='minimal'): bilinear align_corners=True to the multiple of m
    nearest to the target.  CUDA tensors go through the HIP crop-resize kernel; CPU tensors (dataset-side
    preprocessing, tools/test_single_forward.py:20) use torch's interpolate."""

    def __init__(self, width, height, multiple_of=14):
        self.width, self.height, self.m = width, height, multiple_of

    def get_size(self, width, height):
        sh, sw = self.height / height, self.width / width
        nh = int(round(sh * height / self.m) * self.m)
        nw = int(round(sw * width / self.m) * self.m)
        return nw, nh

    def __call__(self, x):
        nw, nh = self.get_size(x.shape[-1], x.shape[-2])
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1:
            from .hip_ops import ops
            H, W = x.shape[-2:]
            out = torch.empty((1, x.shape[1], nh, nw), dtype=torch.float32, device=x.device)
            box = torch.tensor([[0, 0, W, H]], dtype=torch.int32, device=x.device)
            ops.crop_resize(x[0].contiguous(), box, out)
            return out
        return F.interpolate(x, (nh, nw), mode='bilinear', align_corners=True)


def _build_param_tree(root, spec):
    """Register every checkpoint tensor under its dotted name so state_dict()/load_state_dict()/.to()
    behave exactly like the reference's module tree (containers only, no forward code)."""
    for name, e in spec.items():
        parts = name.split('.')
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        leaf = parts[-1]
        if e.dtype.is_floating_point and e.kind not in ("bn_mean", "bn_var", "k_minus_1"):
            # like the reference: branch parameters frozen (patchfusion.py:111-115), fusion-side parameters trainable
            # (so that DistributedDataParallel(model) in tools/test.py:221 accepts the module)
            frozen = name.startswith(("coarse_branch.", "fine_branch."))
            mod.register_parameter(leaf, nn.Parameter(torch.zeros(e.shape, dtype=e.dtype), requires_grad=not frozen))
        else:
            mod.register_buffer(leaf, torch.zeros(e.shape, dtype=e.dtype))


class PatchFusion(nn.Module, PyTorchModelHubMixin):
    def __init__(self, config, compute_dtype=None, ops=None, shard_patches=None, core_providers=None):
        nn.Module.__init__(self)
        # patchfusion.py:64-78: an mmengine ConfigDict (tools/test.py, tools/train.py) forces load_branch=True; any
        # other mapping (the HF `from_pretrained` path hands over a plain dict read from config.json, which was
        # SAVED with load_branch=true and local pretrain_model paths) forces load_branch=False.
        from_mmengine = _is_mmengine_configdict(config)
        if hasattr(config, "to_dict") and not isinstance(config, AttrDict):
            config = config.to_dict()
        config = AttrDict(dict(config))
        config["load_branch"] = bool(from_mmengine)
        if not from_mmengine:
            for br in ("coarse_branch", "fine_branch"):
                if isinstance(config.get(br), dict):
                    config[br]["pretrained_resource"] = None
        self.config = config
        self.min_depth, self.max_depth = config.min_depth, config.max_depth
        self.patch_process_shape = tuple(config.patch_process_shape)
        self.tile_cfg = self.prepare_tile_cfg(config.image_raw_shape, config.patch_split_num)
        self.coarse_branch_cfg = config.coarse_branch
        for br in (config.coarse_branch, config.fine_branch):
            if br.type == 'DA-ZoeDepth':
                if br.midas_model_type not in ('vits', 'vitb', 'vitl'):
                    raise NotImplementedError(br.midas_model_type)
            elif br.type == 'ZoeDepth':
                # MiDaS-core branch (BASELINE configs[4]): the BEiT/MiDaS core lives in an un-vendored torch.hub repo
                # (midas.py:340) and is supplied as a feature provider (engine.ExternalCoreBranchNet, PARITY UNPINNED for
                # the core); the ZoeDepth head, the fusion network and the tiling run on the HIP engine
                from .spec import MIDAS_CORE_CHANNELS
                if br.midas_model_type not in MIDAS_CORE_CHANNELS:
                    raise ValueError(f"Invalid model type: {br.midas_model_type}. Must be one of {list(MIDAS_CORE_CHANNELS)}")
            else:
                raise NotImplementedError
        if config.coarse_branch.type != config.fine_branch.type:
            raise NotImplementedError("coarse and fine branch of different types")
        if config.coarse_branch.bin_centers_type not in ("normed", "softplus", "hybrid1", "hybrid2"):
            raise ValueError("bin_centers_type should be one of 'normed', 'softplus', 'hybrid1', 'hybrid2'")
        # patchfusion.py:82-87: ensure_multiple_of 32 for the MiDaS-core branch, 14 for Depth-Anything
        self.resizer = Resizer(self.patch_process_shape[1], self.patch_process_shape[0],
                               32 if config.coarse_branch.type == 'ZoeDepth' else 14)
        self.core_providers = tuple(core_providers) if core_providers is not None else (None, None)
        self.spec = patchfusion_spec(config)
        _build_param_tree(self, self.spec)
        self.consistency_training = False
        self.compute_dtype = _DTYPES[compute_dtype or config.get("compute_dtype", "fp32")]
        if shard_patches is None:
            shard_patches = bool(config.get("shard_patches", False)) or _os.environ.get("PF_SHARD_PATCHES", "0") == "1"
        self.shard_patches = bool(shard_patches)
        self._ops = ops
        self.overlap_coarse = bool(config.get("overlap_coarse", True)) and _os.environ.get("PF_OVERLAP", "1") != "0"
        self.overlap_batches = bool(config.get("overlap_batches", True)) and _os.environ.get("PF_OVERLAP_BATCHES", "1") != "0"
        self._side_stream = None
        self._aux_streams = []
        self.n_streams = int(_os.environ.get("PF_STREAMS", config.get("n_streams", 2)))
        self.split_single_batch = _os.environ.get("PF_SPLIT_SINGLE_BATCH", "1") != "0"
        # ViT encoder of the fine branch over ALL tiles of this rank in one launch per layer (M = tiles x 1037 token rows) instead of
        # once per process_num batch; the DPT head / fusion keep the process_num batches.  Identical numbers (no op mixes rows).
        # Off by default: measured 290.2 vs 287.0 ms per image (round 3, gpurun_out/r3a_bench_*.json) -- with two streams the tails of
        # the 8-tile launches are already filled by the other batch.
        self.vit_batch_all = bool(config.get("vit_batch_all", False)) or _os.environ.get("PF_VIT_BATCH_ALL", "0") == "1"
        self._engine = None
        self._coarse_state = None
        self._pending_core_sd = [None, None]      # `core.` checkpoint tensors waiting for an external core provider (_load_branch)
        if config.load_branch:
            # patchfusion.py:105-109: each branch checkpoint is loaded with strict=True into its own sub-module
            # BOTH checkpoints pass their strict check before either is applied: a fine-branch checkpoint that fails must not leave the coarse
            # branch's injected core provider already mutated (round-5 advisor finding)
            loaded = [(prefix, torch.load(path, map_location='cpu')['model_state_dict'])
                      for prefix, path in zip(("coarse_branch.", "fine_branch."), config.pretrain_model)]
            for prefix, sd in loaded:
                self._check_branch(prefix, sd)
            for prefix, sd in loaded:
                self._load_branch(prefix, sd)

    def _check_branch(self, prefix, branch_sd):
        """the strict key check of ``_load_branch`` alone (raises; mutates nothing)"""
        want = [k[len(prefix):] for k in self.spec if k.startswith(prefix)]
        wanted = set(want)
        missing = [k for k in want if k not in branch_sd]
        external_core = getattr(self.config[prefix[:-1]], "type", None) == 'ZoeDepth'
        unexpected = [k for k in branch_sd if k not in wanted and not (external_core and k.startswith("core."))]
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict for {prefix[:-1]}: Missing key(s): {missing[:8]}"
                               f"{' ...' if len(missing) > 8 else ''}; Unexpected key(s): {unexpected[:8]}"
                               f"{' ...' if len(unexpected) > 8 else ''}")

    def drop_pending_core_weights(self):
        """forget `core.` checkpoint tensors that are still waiting for a provider able to take them (a full CPU copy of a BEiT core otherwise
        lives as long as the model); also done when the engine is built -- by then the providers in place are the ones that run"""
        self._pending_core_sd = [None, None]

    def _load_branch(self, prefix, branch_sd):
        """``self.<branch>.load_state_dict(sd, strict=True)`` of the reference: missing / unexpected keys raise."""
        want = [k[len(prefix):] for k in self.spec if k.startswith(prefix)]
        missing = [k for k in want if k not in branch_sd]
        wanted = set(want)
        # type 'ZoeDepth': the MiDaS/BEiT core is external (engine.ExternalCoreBranchNet, injected provider); a real branch checkpoint
        # still carries its weights under `core.` -- they belong to the provider, not to this module's parameter tree
        external_core = getattr(self.config[prefix[:-1]], "type", None) == 'ZoeDepth'
        unexpected = [k for k in branch_sd if k not in wanted and not (external_core and k.startswith("core."))]
        # the branch's own strict check comes FIRST: a checkpoint that fails it must not have mutated the injected core (round-4 advisor finding)
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict for {prefix[:-1]}: Missing key(s): {missing[:8]}"
                               f"{' ...' if len(missing) > 8 else ''}; Unexpected key(s): {unexpected[:8]}"
                               f"{' ...' if len(unexpected) > 8 else ''}")
        if external_core:
            # the core's own weights go to the injected provider, strictly, when it can take them (an nn.Module-like provider); a checkpoint that
            # carries `core.` keys with no provider to receive them is reported AND kept: set_core_providers() re-offers it to a provider that
            # arrives later -- the keys are not silently dropped (round-3 / round-4 advisor findings)
            core_sd = {k[len("core."):]: v for k, v in branch_sd.items() if k.startswith("core.") and k not in wanted}
            if core_sd:
                which = 0 if prefix.startswith("coarse") else 1
                provider = self.core_providers[which]
                if provider is not None and hasattr(provider, "load_state_dict"):
                    provider.load_state_dict(core_sd, strict=True)
                else:
                    import warnings
                    self._pending_core_sd[which] = core_sd
                    warnings.warn(f"{prefix[:-1]}: {len(core_sd)} checkpoint tensors under 'core.' belong to the external MiDaS/BEiT core; "
                                  f"{'the injected provider has no load_state_dict' if provider is not None else 'no core provider is set'} "
                                  "-- they are kept and handed to the provider that set_core_providers() installs", stacklevel=2)
            branch_sd = {k: v for k, v in branch_sd.items() if k in wanted}
        return self.load_state_dict({prefix + k: v for k, v in branch_sd.items()}, strict=False)

    # ------------------------------------------------------------------ reference helper surface
    def prepare_tile_cfg(self, image_raw_shape, patch_split_num):
        return tiling.prepare_tile_cfg(self.patch_process_shape, image_raw_shape, patch_split_num)

    def load_dict(self, dict):
        return self.load_state_dict(dict, strict=False)

    def _spec_tensors(self):
        for name in self.spec:
            mod = self
            parts = name.split('.')
            for q in parts[:-1]:
                mod = mod._modules[q]
            t = mod._parameters.get(parts[-1])
            yield name, (t if t is not None else mod._buffers[parts[-1]])

    def load_state_dict(self, state_dict, *a, **k):
        if not getattr(self, "_params_freed", False):
            self._engine = None
            return super().load_state_dict(state_dict, *a, **k)
        # free_parameters() released the checkpoint tensors: the incoming dict has to bring ALL of them back.  A partial dict (strict=False with one
        # branch, the fusion-only `get_save_dict()` format, `_load_branch`) would leave the uncovered tensors at whatever the re-materialisation put
        # there and the next forward would rebuild the engine from them -- wrong depth with no error (round-5 advisor, medium).  Such a call is refused
        # and changes nothing: the engine built from the released tensors keeps serving forward().
        missing = [n for n in self.spec if n not in state_dict]
        if missing:
            raise RuntimeError(f"PatchFusion.free_parameters() released the checkpoint tensors; load_state_dict() must now be given every tensor of the "
                               f"model ({len(missing)} of {len(self.spec)} are missing, e.g. {missing[0]!r}).  Nothing was changed.")
        dev = self._device
        for name, t in self._spec_tensors():
            e = self.spec[name]
            t.data = torch.zeros(e.shape, dtype=e.dtype, device=dev)
        try:
            out = super().load_state_dict(state_dict, *a, **k)
        except Exception:
            for _, t in self._spec_tensors():                  # (a strict / shape error: back to the released state, old engine still valid)
                t.data = torch.empty(0, dtype=t.dtype, device=t.device)
            raise
        self._params_freed = False
        self._engine = None
        return out

    def _apply(self, fn, *a, **k):
        if getattr(self, "_params_freed", False):
            raise RuntimeError("PatchFusion.free_parameters() released the unpacked checkpoint tensors; the module cannot be moved / cast "
                               "(that rebuilds the engine from them).  load_state_dict() the checkpoint again first.")
        self._engine = None
        self._forget_engine_state(release=True)
        return super()._apply(fn, *a, **k)

    def free_parameters(self):
        """OPT-IN memory saver for inference deployments (round-4 review item 7a): once the engine is built, every layer lives in its packed form
        (f32 GEMM layout, split-precision planes, Winograd filters) and the unpacked nn.Parameter / buffer tensors of the two checkpoints are dead
        weight on the device (~3.4 GB for ViT-L; 8x replicated at N = 8).  This builds the engine if needed and releases their storage.  Afterwards
        ``state_dict()`` / ``get_save_dict()`` / ``.to()`` raise (the packed form is not a checkpoint format); ``load_state_dict(sd)`` brings the
        module back to the normal state.  Nothing else changes: forward results are bit-identical."""
        self._ensure_engine()
        for mod in self.modules():
            for t in list(mod._parameters.values()) + list(mod._buffers.values()):
                if t is not None:
                    t.data = torch.empty(0, dtype=t.dtype, device=t.device)
        self._params_freed = True
        return self

    def state_dict(self, *a, **k):
        if getattr(self, "_params_freed", False):
            raise RuntimeError("PatchFusion.free_parameters() released the unpacked checkpoint tensors: there is no state_dict to return "
                               "(load_state_dict() the checkpoint again to restore them)")
        return super().state_dict(*a, **k)

    def _forget_engine_state(self, release=False):
        """library-side state tied to an engine: the cached PF_* switches and per-layer dispatch plans of hip_ops (they reference the packed layers
        of the engine that made them) and, on `release`, the per-stream Winograd arenas (12 + 8 GB per stream at the headline layer: moving the
        module -- .cpu(), .to(other) -- or calling release_workspaces() returns them to the allocator; they re-grow on demand)."""
        import sys
        h = sys.modules.get("patchfusion_amd.hip_ops")
        if h is not None:
            h.refresh_env()
            if release:
                h.release_workspaces()

    @staticmethod
    def release_workspaces():
        """free the engine's per-stream scratch arenas (see hip_ops.release_workspaces); safe at any time between forward calls"""
        import sys
        h = sys.modules.get("patchfusion_amd.hip_ops")
        if h is not None:
            h.release_workspaces()

    def get_save_dict(self):
        return OrderedDict((k, v) for k, v in self.state_dict().items() if 'coarse_branch' not in k and 'fine_branch' not in k)

    def set_core_providers(self, coarse, fine):
        """relative-depth cores of a type-'ZoeDepth' model (see engine.ExternalCoreBranchNet)"""
        self.core_providers = (coarse, fine)
        for which, provider in enumerate(self.core_providers):       # checkpoint `core.` tensors that arrived before the provider did
            pend = self._pending_core_sd[which]
            if pend is not None and provider is not None and hasattr(provider, "load_state_dict"):
                provider.load_state_dict(pend, strict=True)
                self._pending_core_sd[which] = None
        self._engine = None

    def set_compute_dtype(self, dtype):
        self.compute_dtype = _DTYPES[dtype]
        self._engine = None

    # ------------------------------------------------------------------ engine
    @property
    def ops(self):
        if self._ops is None:
            from .hip_ops import ops   # raises if libpf_hip.so is missing: no fallback
            self._ops = ops
        return self._ops

    def _ensure_engine(self):
        if self._engine is None:
            from .engine import BranchNet, ExternalCoreBranchNet, FusionNet, G2LNet
            if getattr(self, "_params_freed", False):
                raise RuntimeError("the engine has to be rebuilt but free_parameters() released the checkpoint tensors: load_state_dict() first")
            sd = self.state_dict()
            dev = next(iter(sd.values())).device
            if self._ops is None and dev.type != "cuda":
                raise RuntimeError("PatchFusion (MI355X engine) needs the model on a GPU: call .cuda() first")
            dt, cfg = self.compute_dtype, self.config
            self._forget_engine_state()                       # PF_* switches are resolved here, once per engine build; cached plans of a previous engine go
            def branch(prefix, bcfg, provider):
                if bcfg.type == 'ZoeDepth':
                    return ExternalCoreBranchNet(sd, prefix, bcfg, self.patch_process_shape, dt, dev, provider)
                return BranchNet(sd, prefix, bcfg, self.patch_process_shape, dt, dev)
            self._engine = dict(
                coarse=branch("coarse_branch.", cfg.coarse_branch, self.core_providers[0]),
                fine=branch("fine_branch.", cfg.fine_branch, self.core_providers[1]),
                g2l=G2LNet(sd, cfg.guided_fusion, dt, dev),
                fusion=FusionNet(sd, cfg, dt, dev))
            self._device = dev
            self._mask_cache = {}
            self._table_cache = {}
            self.drop_pending_core_weights()                  # the providers in place are the ones that run
        return self._engine

    def _mask(self, shape):
        key = tuple(shape)
        if key not in self._mask_cache:
            m = torch.from_numpy(tiling.gaussian_blend_mask(key) + 1e-3).to(self._device)
            self._mask_cache[key] = m.contiguous()
        return self._mask_cache[key]

    # ------------------------------------------------------------------ branch-level API
    @torch.no_grad()
    def _coarse(self, image_lr, taps=None):
        nets = self._ensure_engine()
        depth, feats = nets["coarse"].forward(self.ops, image_lr.contiguous().float(), taps)
        g2l = nets["g2l"].forward(self.ops, feats)        # patch invariant -> once per image (exact)
        self._coarse_state = dict(depth=depth.view(depth.shape[0], 1, *depth.shape[1:]), feats=feats, g2l=g2l)
        return self._coarse_state

    def coarse_forward(self, image_lr):
        """-> (coarse_prediction [1,1,h,w] f32, six coarse feature maps NCHW f32) like patchfusion.py:189-206"""
        st = self._coarse(image_lr)
        feats = [self.ops.nhwc_to_nchw(f) for f in st["feats"]]
        # the tensors handed to the caller are kept alive and remembered by IDENTITY and version counter: `infer_forward(tile_temp=...)`
        # reuses the engine's own state only for exactly these objects, unmodified (an address comparison would also match foreign
        # tensors that landed on recycled addresses, or the same tensors after an in-place write)
        st["handed_out"] = tuple(feats) + (st["depth"],)
        st["handed_out_versions"] = tuple(t._version for t in st["handed_out"])
        return st["depth"], feats

    def fine_forward(self, image_hr_crop):
        nets = self._ensure_engine()
        depth, feats = nets["fine"].forward(self.ops, image_hr_crop.contiguous().float())
        return depth.unsqueeze(1), [self.ops.nhwc_to_nchw(f) for f in feats]

    def _state_from_tile_temp(self, tile_temp):
        """The coarse state handed over by a caller that follows the reference protocol (patchfusion.py:410-414:
        ``tile_temp = {'coarse_prediction': [1,1,h,w], 'coarse_features': [6 x NCHW]}``).  When these are the tensors the last
        ``coarse_forward`` returned, the engine's own NHWC state (and the G2L maps derived from it) is reused; foreign
        tensors are re-laid out (a permute -- no arithmetic) and the patch-invariant G2L stacks are recomputed from them."""
        st = self._coarse_state
        feats = tile_temp['coarse_features']
        given = tuple(feats) + (tile_temp['coarse_prediction'],)
        mine = st.get("handed_out") if st is not None else None
        if (mine is not None and len(mine) == len(given) and all(a is b for a, b in zip(mine, given))
                and tuple(t._version for t in given) == st["handed_out_versions"]):
            return st
        nets = self._ensure_engine()
        dev, dt = self._device, self.compute_dtype
        nhwc = [f.detach().to(dev).permute(0, 2, 3, 1).contiguous().to(dt) for f in feats]
        depth = tile_temp['coarse_prediction'].detach().to(device=dev, dtype=torch.float32).contiguous()
        st = dict(depth=depth, feats=nhwc, g2l=nets["g2l"].forward(self.ops, nhwc), handed_out=given,
                  handed_out_versions=tuple(t._version for t in given))
        self._coarse_state = st
        return st

    @torch.no_grad()
    def infer_forward(self, imgs_crop, bbox_feat_forward, tile_temp=None, coarse_temp_dict=None, taps=None):
        """fine branch + fusion for one batch of crops (patchfusion.py:343-356).  ``bbox_feat_forward``
        [B,5] = (0, x1,y1,x2,y2) in process coordinates.  ``tile_temp`` (``coarse_prediction`` + ``coarse_features``,
        the way baseline_pretrain.py:293-307 passes them) selects the coarse state; without it the state of the last
        ``coarse_forward`` / ``forward`` call is used.  ``coarse_temp_dict`` (the P-times repeated ROI crops of the
        reference, patchfusion.py:240-257) is accepted and not read: the same ROI values are gathered from the coarse
        maps inside the kernels from ``bbox_feat_forward`` -- identical numbers, no 1 GB transient."""
        nets = self._ensure_engine()
        st = self._state_from_tile_temp(tile_temp) if tile_temp is not None else self._coarse_state
        assert st is not None, "run coarse_forward first"
        crops = imgs_crop.contiguous().float()
        rois = bbox_feat_forward
        if rois.device != crops.device or rois.dtype != torch.float32 or not rois.is_contiguous():
            rois = rois.to(device=crops.device, dtype=torch.float32).contiguous()
        fdepth, ffeats = nets["fine"].forward(self.ops, crops)
        d = nets["fusion"].forward(self.ops, crops, rois, fdepth, ffeats, st["depth"], st["feats"], st["g2l"], taps)
        return d.unsqueeze(1)

    # ------------------------------------------------------------------ tiles
    def _rois(self, boxes, tile_cfg, device):
        """bboxs * bboxs_feat_factor in float32 exactly like baseline_pretrain.py:275-282 (int32 boxes times
        a float32 factor tensor), batch index 0."""
        H, W = tile_cfg['image_raw_shape']
        ps = self.patch_process_shape
        fac = torch.tensor([1 / W * ps[1], 1 / H * ps[0], 1 / W * ps[1], 1 / H * ps[0]]).unsqueeze(0)
        bf = torch.tensor(boxes, dtype=torch.int32) * fac
        return torch.cat([torch.zeros(len(boxes), 1), bf], dim=-1).to(device).contiguous()

    def _tile_tables(self, tiles, tile_cfg):
        """Device-resident integer boxes and float ROIs of ALL tiles, uploaded in ONE host->device copy per
        forward (cached for the deterministic m1/m2 schedules): a pageable H2D copy inside the batch loop
        would synchronise the null stream and drain the launch queue before every batch."""
        key = (tuple(t['box'] for t in tiles), tuple(tile_cfg['image_raw_shape']))
        hit = self._table_cache.get(key)
        if hit is None:
            boxes = [t['box'] for t in tiles]
            bt = torch.tensor(boxes, dtype=torch.int32).to(self._device)
            rois = self._rois(boxes, tile_cfg, self._device)
            hit = (bt, rois)
            if len(self._table_cache) > 8:
                self._table_cache.clear()
            self._table_cache[key] = hit
        return hit

    def _sharding(self):
        """(rank, world) of the single-image tile sharding; (0, 1) unless it was explicitly requested."""
        if self.shard_patches and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_rank(), torch.distributed.get_world_size()
        return 0, 1

    def _same_image_token(self, image_lr, image_hr):
        """Tile sharding is only meaningful when every rank holds the SAME image (tools/test.py's DistributedSampler
        gives every rank a different one).  One all_reduce(MAX) of [c, -c] for a 4-number checksum of the inputs
        proves equality (max == min on every component); a mismatch raises on ALL ranks, so nobody is left waiting in
        the later all_gather."""
        import torch.distributed as dist
        c = torch.stack([image_hr.float().sum(), image_hr.float().abs().max(), image_hr.float()[..., ::97].sum(),
                         image_lr.float().sum()]).double()
        c = torch.cat([c, c.new_full((1,), float(image_hr.shape[-2])), c.new_full((1,), float(image_hr.shape[-1]))])  # no H2D copy
        v = torch.cat([c, -c])
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        n = c.numel()
        return (v[:n] == -v[n:]).all()      # device bool; read (one host sync) only after all tile kernels are queued

    @staticmethod
    def _raise_if_images_differ(same, world):
        if same is not None and not bool(same):
            raise RuntimeError(
                "PatchFusion(shard_patches=True): the ranks hold DIFFERENT images -- tile sharding needs the same image on "
                f"all {world} ranks.  For image-level data parallelism (tools/test.py / dist_test.sh with a "
                "DistributedSampler) leave shard_patches off (the default).")

    @staticmethod
    def _broadcast_schedule(tiles, world):
        """r<N>: tile positions come from python's process-global `random`; every rank must paste the SAME tiles,
        so rank 0's schedule is authoritative (a few hundred ints)."""
        import torch.distributed as dist
        box = [tiles]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    @torch.no_grad()
    def _predict_tiles(self, image_hr, tiles, tile_cfg, process_num, coarse_ready=None, same_image=None):
        """Per-tile depth [P,h,w] f32 for this rank's shard (all tiles when not distributed)."""
        ops, dev = self.ops, self._device
        ph, pw = self.patch_process_shape
        n = len(tiles)
        rank, world = self._sharding()
        lo, hi = tiling.shard_range(n, rank, world)
        preds = ops.empty((n, ph, pw), torch.float32, dev)
        img = image_hr[0].contiguous().float()
        bt, rois = self._tile_tables(tiles, tile_cfg)
        nets = self._engine
        # consecutive batches alternate between the current stream and an auxiliary stream: the low-occupancy
        # kernels of one batch (coarse pyramid levels L0..L2, B x 14x19 ... 56x74 maps) fill the gaps of the other
        # A shard that fits ONE process_num batch (the multi-GPU case: P/N <= process_num tiles per rank) is cut in two half batches so the
        # same overlap applies (identical numbers: no op mixes rows of a batch).  PF_SPLIT_SINGLE_BATCH=0 keeps the single batch.
        if (img.is_cuda and self.overlap_batches and self.split_single_batch and 4 <= (hi - lo) <= process_num):
            process_num = (hi - lo + 1) // 2
        use_aux = img.is_cuda and self.overlap_batches and (hi - lo) > process_num
        main = torch.cuda.current_stream() if img.is_cuda else None
        streams = [main]
        if use_aux:
            while len(self._aux_streams) < self.n_streams - 1:
                self._aux_streams.append(torch.cuda.Stream(device=dev))
            streams += self._aux_streams[:self.n_streams - 1]
            for a in streams[1:]:
                a.wait_stream(main)
        crops_all = vit_all = None
        if self.vit_batch_all and (hi - lo) > process_num and hasattr(nets["fine"], "vit"):
            crops_all = ops.empty((hi - lo, 3, ph, pw), torch.float32, dev)
            ops.crop_resize(img, bt[lo:hi], crops_all)
            vit_all = nets["fine"].vit(ops, crops_all)          # 4 x [tiles, th, tw, D]; runs while coarse + G2L use the side stream
            for a in streams[1:]:
                a.wait_stream(main)
        for bi, s in enumerate(range(lo, hi, process_num)):
            e = min(s + process_num, hi)
            stream = streams[bi % len(streams)]
            ctx = torch.cuda.stream(stream) if img.is_cuda else _nullcontext()
            with ctx:
                if crops_all is not None:
                    crops = crops_all[s - lo:e - lo]
                    fdepth, ffeats = nets["fine"].forward(ops, crops, vit_feats=[f[s - lo:e - lo] for f in vit_all])
                else:
                    crops = ops.empty((e - s, 3, ph, pw), torch.float32, dev)
                    ops.crop_resize(img, bt[s:e], crops)
                    # the fine branch does not depend on the coarse pass: it runs while the coarse branch + G2L
                    # (batch 1, low occupancy) execute on the side stream
                    fdepth, ffeats = nets["fine"].forward(ops, crops)
                if coarse_ready is not None:
                    stream.wait_event(coarse_ready)
                st = self._coarse_state
                d = nets["fusion"].forward(ops, crops, rois[s:e], fdepth, ffeats, st["depth"], st["feats"], st["g2l"])
                ops.copy_plane(d.unsqueeze(1), preds[s:e])
        for a in streams[1:]:
            main.wait_stream(a)
        if world > 1:
            from .dist import all_gather_shards
            self._raise_if_images_differ(same_image, world)   # every rank sees the same verdict -> all raise or none
            preds = all_gather_shards(preds, n, world)
        return preds

    def _paste_table(self, paste):
        hit = self._table_cache.get(('paste', paste))
        if hit is None:
            hit = torch.tensor(list(paste), dtype=torch.int32).to(self._device)
            self._table_cache[('paste', paste)] = hit
        return hit

    def _stitch(self, preds, tiles, tile_cfg, cai_mode='m1'):
        ops, dev = self.ops, self._device
        ph, pw = self.patch_process_shape
        RH, RW = tile_cfg['patch_reensemble_shape']
        mask = self._mask((ph, pw))
        init = [i for i, t in enumerate(tiles) if t['phase'] == 'init']
        pred = ops.zeros((RH, RW), torch.float32, dev)
        count = ops.zeros((RH, RW), torch.float32, dev)
        yx = self._paste_table(tuple(tiles[i]['paste'] for i in init))
        ops.stitch_init(pred, count, preds[init[0]:init[-1] + 1].contiguous(), mask, yx)
        avg = ops.empty((RH, RW), torch.float32, dev)
        ops.stitch_finish_init(avg, pred, count)
        resized = False
        for i, t in enumerate(tiles):
            if t['phase'] == 'regular':
                ops.stitch_update(avg, count, preds[i], mask, t['paste'][0], t['paste'][1])
            elif t['phase'] == 'random':
                if not resized:   # RunningAverageMap.resize (utils.py:32-36): avg nearest, count bilinear
                    H, W = tile_cfg['image_raw_shape']
                    a2, c2 = ops.empty((H, W), torch.float32, dev), ops.empty((H, W), torch.float32, dev)
                    ops.resize_nearest_f32(avg, a2)
                    ops.resize_bilinear_f32(count, c2)
                    avg, count = a2, c2
                    raw_mask = self._mask(tile_cfg['patch_raw_shape'])
                    resized = True
                ops.stitch_update(avg, count, preds[i], raw_mask, t['paste'][0], t['paste'][1])
        if cai_mode[0] == 'r' and not resized:
            # r<N> that draws zero random tiles: the reference still runs avg_depth_map.resize(image_raw_shape)
            # (patchfusion.py:436-438, baseline_pretrain.py:401-403) -> nearest resize of the map to the raw resolution
            H, W = tile_cfg['image_raw_shape']
            a2 = ops.empty((H, W), torch.float32, dev)
            ops.resize_nearest_f32(avg, a2)
            avg = a2
        return avg

    # ------------------------------------------------------------------ training-mode forward (forward VALUE only)
    @torch.no_grad()
    def train_forward(self, image_lr, crops_image_hr, crop_depths, bboxs):
        """``forward(mode='train')`` of the reference, patchfusion.py:372-399: B images, one random crop
        each -> coarse branch on ``image_lr`` [B,3,h,w], fine branch on ``crops_image_hr`` [B,3,h,w],
        ``coarse_postprocess_train`` (roi_align of image i's coarse maps with box i, :227-237), ``fusion_forward``, SILogLoss
        against ``crop_depths`` [B,1,h,w].  Returns ``(loss_dict, {'rgb','depth_pred','depth_gt'})`` like the reference.
        FORWARD VALUE ONLY: the engine has no backward kernels, the returned tensors carry no autograd graph (use it for
        validation loss / to check a training pipeline's data path; optimisation steps need the reference's PyTorch modules)."""
        nets = self._ensure_engine()
        ops, dev = self.ops, self._device
        B = image_lr.shape[0]
        assert crops_image_hr.shape[0] == B and bboxs.shape[0] == B
        H, W = self.tile_cfg['image_raw_shape']
        ps = self.patch_process_shape
        # bboxs * bboxs_feat_factor in float32 exactly like patchfusion.py:373-380, batch index i = image i
        fac = torch.tensor([1 / W * ps[1], 1 / H * ps[0], 1 / W * ps[1], 1 / H * ps[0]], device=bboxs.device).unsqueeze(0)
        bf = bboxs * fac
        rois = torch.cat((torch.arange(B, device=bboxs.device).unsqueeze(-1), bf), dim=-1).to(device=dev, dtype=torch.float32).contiguous()
        cdepth, cfeats = nets["coarse"].forward(ops, image_lr.contiguous().float())
        g2l = nets["g2l"].forward(ops, cfeats)            # per image here (batch B): the coarse maps differ per sample
        crops = crops_image_hr.contiguous().float()
        fdepth, ffeats = nets["fine"].forward(ops, crops)
        d = nets["fusion"].forward(ops, crops, rois, fdepth, ffeats, cdepth.view(B, 1, *cdepth.shape[1:]), cfeats, g2l)
        depth_prediction = d.unsqueeze(1)
        loss_dict = {}
        if crop_depths is not None:
            loss_dict['sig_loss'] = self._sigloss(depth_prediction, crop_depths, self.config.get('sigloss'))
            loss_dict['total_loss'] = loss_dict['sig_loss']
        return loss_dict, {'rgb': crops_image_hr, 'depth_pred': depth_prediction, 'depth_gt': crop_depths}

    def _sigloss(self, pred, target, sigloss_cfg=None):
        """SILogLoss.forward (losses.py:15-62) on the device: bilinear align_corners resize of the prediction when the grids differ
        (:27-28), then the masked scale-invariant log loss (pf_silog_loss); beta from the loss config (default 0.15)."""
        gt = target.to(device=self._device, dtype=torch.float32)
        if tuple(gt.shape[-2:]) != tuple(pred.shape[-2:]):
            pred = F.interpolate(pred, tuple(gt.shape[-2:]), mode='bilinear', align_corners=True)
        beta = float(sigloss_cfg.get('beta', 0.15)) if isinstance(sigloss_cfg, dict) else 0.15
        return self.ops.silog_loss(pred.contiguous(), gt.contiguous(), self.min_depth, self.max_depth, beta)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, mode, image_lr, image_hr, depth_gt=None, crops_image_hr=None, crop_depths=None, bboxs=None,
                tile_cfg=None, cai_mode='m1', process_num=4):
        if mode == 'train':
            return self.train_forward(image_lr, crops_image_hr, crop_depths, bboxs)
        if tile_cfg is None:
            tile_cfg = self.tile_cfg
        else:
            tile_cfg = self.prepare_tile_cfg(tile_cfg['image_raw_shape'], tile_cfg['patch_split_num'])
        assert image_hr.shape[0] == 1
        self._ensure_engine()
        coarse_ready = None
        if image_hr.is_cuda and self.overlap_coarse:
            # coarse branch + G2L on a side stream, overlapped with the first fine-branch batch
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=image_hr.device)
            main = torch.cuda.current_stream()
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                self._coarse(image_lr)
                coarse_ready = torch.cuda.Event()
                coarse_ready.record(self._side_stream)
        else:
            self._coarse(image_lr)
        tiles = tiling.tile_schedule(tile_cfg, self.patch_process_shape, cai_mode, process_num)
        _, world = self._sharding()
        same = None
        if world > 1:
            same = self._same_image_token(image_lr, image_hr)
            if cai_mode[0] == 'r':
                tiles = self._broadcast_schedule(tiles, world)
        preds = self._predict_tiles(image_hr, tiles, tile_cfg, process_num, coarse_ready, same)
        avg = self._stitch(preds, tiles, tile_cfg, cai_mode)
        depth = avg.unsqueeze(0).unsqueeze(0)
        return depth, {'rgb': image_lr, 'depth_pred': depth, 'depth_gt': depth_gt}
