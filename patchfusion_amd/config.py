"""Model-config helpers for the PatchFusion boundary.

``make_config`` reproduces the ``model.config`` section of the reference's
configs/patchfusion_depthanything/depthanything_{vits,vitb,vitl}_patchfusion_u4k.py:71-90 for an
arbitrary process shape (the reference uses 392x518; tests also use smaller multiples of 14).
``AttrDict`` is the attribute-accessible nested mapping the reference expects of config
sections (patchfusion.py:77-90 uses both ``config.coarse_branch.type`` and ``**config.coarse_branch``).
"""
from .spec import DPT_ARCH, VIT_PATCH


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict.wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict.wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [plain(x) for x in v]
            return v
        return plain(self)

    # the reference's `config` is a transformers.PretrainedConfig: tools/convert_huggingface.py:79 calls
    # model.config.to_json_file(...) next to save_pretrained; keep that part of the surface
    def to_json_string(self):
        import json
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def to_json_file(self, json_file_path):
        with open(json_file_path, "w", encoding="utf-8") as f:
            f.write(self.to_json_string())


def pyramid_sizes(process_shape, branch_type="DA-ZoeDepth"):
    """(h, w) of the six feature levels L5..L0 at ``process_shape``.
    Depth-Anything branch (external/depth_anything/dpt.py:41-63,97-130): full res, 8x/4x/2x/1x the token grid and the
    stride-2 conv of the token grid.  MiDaS-core ZoeDepth branch (guided_fusion_model.py:112 defaults, 384x512 ->
    384x512, 192x256, 96x128, 48x64, 24x32, 12x16): full res and the /2 ... /32 levels of the DPT decoder."""
    h, w = process_shape
    if branch_type == "ZoeDepth":
        return [(h, w)] + [(h // s, w // s) for s in (2, 4, 8, 16, 32)]
    th, tw = h // VIT_PATCH, w // VIT_PATCH
    return [(h, w), (th * 8, tw * 8), (th * 4, tw * 4), (th * 2, tw * 2), (th, tw), ((th + 1) // 2, (tw + 1) // 2)]


def zoe_branch_config(encoder, process_shape, min_depth=1e-3, max_depth=80):
    return dict(
        type="DA-ZoeDepth", min_depth=min_depth, max_depth=max_depth, depth_anything=True,
        midas_model_type=encoder, img_size=list(process_shape), pretrained_resource=None,
        use_pretrained_midas=True, train_midas=True, freeze_midas_bn=True, do_resize=False,
        attractor_alpha=1000, attractor_gamma=2, attractor_kind="mean", attractor_type="inv",
        bin_centers_type="softplus", bin_embedding_dim=128, inverse_midas=False, max_temp=50.0,
        min_temp=0.0212, memory_efficient=True, n_attractors=[16, 8, 4, 1], n_bins=64,
        output_distribution="logbinomial", force_keep_ar=True)


def zoe_midas_branch_config(process_shape=(384, 512), min_depth=1e-3, max_depth=80):
    """the branch section of configs/patchfusion_zoedepth/zoedepth_patchfusion_u4k.py:10-50 (MiDaS DPT_BEiT_L_384 core)"""
    c = zoe_branch_config("vitl", process_shape, min_depth, max_depth)
    c.update(type="ZoeDepth", midas_model_type="DPT_BEiT_L_384", pretrained_resource=None)
    c.pop("depth_anything")
    return c


def make_zoe_config(process_shape=(384, 512), image_raw_shape=(2160, 3840), patch_split_num=(4, 4), min_depth=1e-3, max_depth=80,
                    explicit_fusion_geometry=None):
    """``model.config`` of configs/patchfusion_zoedepth/zoedepth_patchfusion_u4k.py:55-72 (BASELINE configs[4]).  The shipped
    file relies on GuidedFusionPatchFusion's 384x512 defaults; for other (test) process shapes -- multiples of 32 -- the
    geometry is spelled out like the Depth-Anything configs do."""
    gf = dict(type="GuidedFusionPatchFusion", n_channels=5, g2l=True)
    if explicit_fusion_geometry or tuple(process_shape) != (384, 512):
        sizes = pyramid_sizes(process_shape, "ZoeDepth")
        gf.update(patch_process_shape=tuple(process_shape), in_channels=[32, 256, 256, 256, 256, 256],
                  num_patches=[a * b for a, b in sizes])
    return dict(
        image_raw_shape=tuple(image_raw_shape), patch_split_num=tuple(patch_split_num),
        patch_process_shape=tuple(process_shape), min_depth=min_depth, max_depth=max_depth,
        load_branch=False, pretrain_model=["", ""],
        coarse_branch=zoe_midas_branch_config(process_shape, min_depth, max_depth),
        fine_branch=zoe_midas_branch_config(process_shape, min_depth, max_depth),
        guided_fusion=gf, sigloss=dict(type="SILogLoss"))


def make_config(encoder="vitl", process_shape=(392, 518), image_raw_shape=(2160, 3840),
                patch_split_num=(4, 4), min_depth=1e-3, max_depth=80):
    C = DPT_ARCH[encoder][0]
    sizes = pyramid_sizes(process_shape)
    return dict(
        image_raw_shape=tuple(image_raw_shape), patch_split_num=tuple(patch_split_num),
        patch_process_shape=tuple(process_shape), min_depth=min_depth, max_depth=max_depth,
        load_branch=False, pretrain_model=["", ""],
        coarse_branch=zoe_branch_config(encoder, process_shape, min_depth, max_depth),
        fine_branch=zoe_branch_config(encoder, process_shape, min_depth, max_depth),
        guided_fusion=dict(type="GuidedFusionPatchFusion", patch_process_shape=tuple(process_shape),
                           in_channels=[32, C, C, C, C, C], num_patches=[a * b for a, b in sizes],
                           n_channels=5, g2l=True),
        sigloss=dict(type="SILogLoss"))
