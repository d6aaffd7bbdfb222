"""Parameter specification of the PatchFusion checkpoint surface.

The drop-in contract (SURVEY.md section 8b "Weights") is that reference checkpoints load by key:
``coarse_branch.*``, ``fine_branch.*`` and the fusion-side keys.  This module derives every
(key, shape, dtype) programmatically from the model config, in the order the reference's
``state_dict()`` yields them:

  estimator/models/patchfusion.py:57-173                (PatchFusion.__init__)
  external/zoedepth/models/zoedepth/zoedepth_v1.py:39-123 (ZoeDepth head)
  external/depth_anything/dpt.py:20-95,133-144          (DPTHead, DPT_DINOv2)
  external/torchhub/.../vision_transformer.py:44-172    (DinoVisionTransformer)
  estimator/models/blocks/guided_fusion_model.py:103-161, swin_layers.py:85-410

Nothing here executes arithmetic of the hot path; it is the schema the weight packer
(packing.py) and the boundary module (patchfusion.py) are built from.
"""
from collections import OrderedDict, namedtuple

import torch

Entry = namedtuple("Entry", "shape dtype kind")

VIT_ARCH = {
    # encoder: (embed dim, depth, heads)        vision_transformer.py:339-378
    "vits": (384, 12, 6),
    "vitb": (768, 12, 12),
    "vitl": (1024, 24, 16),
}
DPT_ARCH = {
    # encoder: (features, out_channels)          zoedepth/models/base_models/depth_anything.py:346-352
    "vits": (64, [48, 96, 192, 384]),
    "vitb": (128, [96, 192, 384, 768]),
    "vitl": (256, [256, 512, 1024, 1024]),
}
N_MIDAS_OUT = 32          # zoedepth_v1.py:84, patchfusion.py:119
VIT_PATCH = 14
VIT_PRETRAIN_GRID = 37    # hubconf img_size=518 -> 37x37 (+cls) stored pos_embed
G2L_WINDOW = 12           # guided_fusion_model.py:140
G2L_DEPTH = [2, 2, 3, 3, 4, 4]          # guided_fusion_model.py:109 (defaults; reversed at use)
G2L_HEADS = [8, 8, 16, 16, 32, 32]      # guided_fusion_model.py:110


def _w(shape):
    return Entry(tuple(shape), torch.float32, "w")


def _b(n):
    return Entry((n,), torch.float32, "b")


def _conv(d, name, cout, cin, k, bias=True):
    d[name + ".weight"] = _w((cout, cin, k, k))
    if bias:
        d[name + ".bias"] = _b(cout)


def _linear(d, name, cout, cin):
    d[name + ".weight"] = _w((cout, cin))
    d[name + ".bias"] = _b(cout)


def _ln(d, name, n):
    d[name + ".weight"] = Entry((n,), torch.float32, "ln_w")
    d[name + ".bias"] = Entry((n,), torch.float32, "ln_b")


def _bn(d, name, n):
    d[name + ".weight"] = Entry((n,), torch.float32, "ln_w")
    d[name + ".bias"] = Entry((n,), torch.float32, "ln_b")
    d[name + ".running_mean"] = Entry((n,), torch.float32, "bn_mean")
    d[name + ".running_var"] = Entry((n,), torch.float32, "bn_var")
    d[name + ".num_batches_tracked"] = Entry((), torch.int64, "zero_int")


def vit_spec(d, p, enc):
    D, depth, _ = VIT_ARCH[enc]
    d[p + "cls_token"] = Entry((1, 1, D), torch.float32, "cls")
    d[p + "pos_embed"] = Entry((1, VIT_PRETRAIN_GRID * VIT_PRETRAIN_GRID + 1, D), torch.float32, "pos")
    d[p + "mask_token"] = Entry((1, D), torch.float32, "cls")
    _conv(d, p + "patch_embed.proj", D, 3, VIT_PATCH)
    for i in range(depth):
        b = f"{p}blocks.{i}."
        _ln(d, b + "norm1", D)
        _linear(d, b + "attn.qkv", 3 * D, D)
        _linear(d, b + "attn.proj", D, D)
        d[b + "ls1.gamma"] = Entry((D,), torch.float32, "gamma")
        _ln(d, b + "norm2", D)
        _linear(d, b + "mlp.fc1", 4 * D, D)
        _linear(d, b + "mlp.fc2", D, 4 * D)
        d[b + "ls2.gamma"] = Entry((D,), torch.float32, "gamma")
    _ln(d, p + "norm", D)


def dpt_head_spec(d, p, enc):
    D = VIT_ARCH[enc][0]
    C, oc = DPT_ARCH[enc]
    for i in range(4):
        _conv(d, f"{p}projects.{i}", oc[i], D, 1)
    # ConvTranspose2d weights are [in, out, k, k]
    d[p + "resize_layers.0.weight"] = _w((oc[0], oc[0], 4, 4))
    d[p + "resize_layers.0.bias"] = _b(oc[0])
    d[p + "resize_layers.1.weight"] = _w((oc[1], oc[1], 2, 2))
    d[p + "resize_layers.1.bias"] = _b(oc[1])
    _conv(d, p + "resize_layers.3", oc[3], oc[3], 3)
    for i in range(4):
        _conv(d, f"{p}scratch.layer{i + 1}_rn", C, oc[i], 3, bias=False)
    for i in range(1, 5):
        r = f"{p}scratch.refinenet{i}."
        _conv(d, r + "out_conv", C, C, 1)
        for u in (1, 2):
            _conv(d, f"{r}resConfUnit{u}.conv1", C, C, 3)
            _conv(d, f"{r}resConfUnit{u}.conv2", C, C, 3)
    _conv(d, p + "scratch.output_conv1", C // 2, C, 3)
    _conv(d, p + "scratch.output_conv2.0", N_MIDAS_OUT, C // 2, 3)
    _conv(d, p + "scratch.output_conv2.2", 1, N_MIDAS_OUT, 1)


def bins_head_spec(d, p, C, n_bins, emb, n_attractors, bin_centers_type="softplus"):
    """The metric-bins head; shared by ZoeDepth (zoedepth_v1.py:84-123) and the fusion head
    (patchfusion.py:149-170).  The bounded attractor layer ('normed' / 'hybrid2', attractor.py:81) has 2 x n_attractors outputs."""
    amul = 2 if bin_centers_type in ("normed", "hybrid2") else 1
    _conv(d, p + "seed_bin_regressor._net.0", 256, C, 1)
    _conv(d, p + "seed_bin_regressor._net.2", n_bins, 256, 1)
    _conv(d, p + "seed_projector._net.0", 128, C, 1)
    _conv(d, p + "seed_projector._net.2", emb, 128, 1)
    for i in range(4):
        _conv(d, f"{p}projectors.{i}._net.0", 128, C, 1)
        _conv(d, f"{p}projectors.{i}._net.2", emb, 128, 1)
    for i in range(4):
        _conv(d, f"{p}attractors.{i}._net.0", 128, emb, 1)
        _conv(d, f"{p}attractors.{i}._net.2", n_attractors[i] * amul, 128, 1)
    d[p + "conditional_log_binomial.log_binomial_transform.k_idx"] = Entry((1, n_bins, 1, 1), torch.int64, "k_idx")
    d[p + "conditional_log_binomial.log_binomial_transform.K_minus_1"] = Entry((1, 1, 1, 1), torch.float32, "k_minus_1")
    cin = N_MIDAS_OUT + 1 + emb
    _conv(d, p + "conditional_log_binomial.mlp.0", cin // 2, cin, 1)
    _conv(d, p + "conditional_log_binomial.mlp.2", 4, cin // 2, 1)


MIDAS_CORE_CHANNELS = {
    # external/zoedepth/models/base_models/midas.py:368-376 (MIDAS_SETTINGS): btlnck + 4 decoder levels
    "DPT_BEiT_L_384": 256, "DPT_BEiT_L_512": 256, "DPT_BEiT_B_384": 256, "DPT_SwinV2_L_384": 256, "DPT_SwinV2_B_384": 256,
    "DPT_SwinV2_T_256": 256, "DPT_Large": 256, "DPT_Hybrid": 256,
}


def branch_channels(bcfg):
    """feature channels C of a branch's five decoder levels (the sixth, `out_conv`, always has N_MIDAS_OUT = 32)"""
    if bcfg["type"] == "DA-ZoeDepth":
        return DPT_ARCH[bcfg["midas_model_type"]][0]
    return MIDAS_CORE_CHANNELS[bcfg["midas_model_type"]]


def branch_spec(d, p, bcfg):
    """type 'DA-ZoeDepth': DINOv2 ViT + DPT core + ZoeDepth head.  type 'ZoeDepth' (MiDaS core, BASELINE configs[4]): only
    the ZoeDepth head is specified here -- the MiDaS/BEiT core lives in an un-vendored torch.hub repository (midas.py:340) and is
    supplied by a feature provider (engine.ExternalCoreBranchNet); its `core.*` checkpoint keys are not part of the schema."""
    C = branch_channels(bcfg)
    if bcfg["type"] == "DA-ZoeDepth":
        enc = bcfg["midas_model_type"]
        vit_spec(d, p + "core.core.pretrained.", enc)
        dpt_head_spec(d, p + "core.core.depth_head.", enc)
    _conv(d, p + "conv2", C, C, 1)
    bins_head_spec(d, p, C, bcfg["n_bins"], bcfg["bin_embedding_dim"], bcfg["n_attractors"], bcfg.get("bin_centers_type", "softplus"))


GF_DEFAULT_IN_CHANNELS = [32, 256, 256, 256, 256, 256]                      # guided_fusion_model.py:108
GF_DEFAULT_NUM_PATCHES = [384 * 512, 192 * 256, 96 * 128, 48 * 64, 24 * 32, 12 * 16]  # guided_fusion_model.py:112


def guided_fusion_spec(d, p, gcfg):
    ch = list(gcfg.get("in_channels", GF_DEFAULT_IN_CHANNELS))                      # [32, C, C, C, C, C]
    n_in = gcfg.get("n_channels", 5)
    num_patches = list(gcfg.get("num_patches", GF_DEFAULT_NUM_PATCHES))
    depth = list(gcfg.get("depth", G2L_DEPTH))
    heads = list(gcfg.get("num_heads", G2L_HEADS))

    def double_conv_bn(name, cin, cout):
        _conv(d, name + ".double_conv.0", cout, cin, 3, bias=False)
        _bn(d, name + ".double_conv.1", cout)
        _conv(d, name + ".double_conv.3", cout, cout, 3, bias=False)
        _bn(d, name + ".double_conv.4", cout)

    def double_conv_wobn(name, cin, cmid, cout):
        _conv(d, name + ".double_conv.0", cmid, cin, 3)
        _conv(d, name + ".double_conv.2", cout, cmid, 3)

    double_conv_bn(p + "inc", n_in, ch[0])
    for i in range(len(ch) - 1):
        double_conv_bn(f"{p}down_conv_list.{i}.maxpool_conv.1", ch[i], ch[i + 1])
    inv = ch[::-1]
    for i in range(1, len(ch)):
        cin = inv[i] + 2 * inv[i - 1]
        double_conv_wobn(f"{p}up_conv_list.{i - 1}.conv", cin, cin, inv[i])
    if gcfg.get("g2l", True):
        heads_inv, depth_inv, np_inv = heads[::-1], depth[::-1], num_patches[::-1]
        nrel = (2 * G2L_WINDOW - 1) ** 2
        for i in range(len(inv)):
            C = inv[i]
            g = f"{p}g2l_list.{i}."
            d[g + "absolute_pos_embed"] = Entry((1, np_inv[i], C), torch.float32, "pos")
            for j in range(depth_inv[i]):
                b = f"{g}g2l_layer.blocks.{j}."
                _ln(d, b + "norm1", C)
                d[b + "attn.relative_position_bias_table"] = Entry((nrel, heads_inv[i]), torch.float32, "relpos")
                d[b + "attn.relative_position_index"] = Entry((G2L_WINDOW ** 2, G2L_WINDOW ** 2), torch.int64, "relidx")
                _linear(d, b + "attn.qkv", 3 * C, C)
                _linear(d, b + "attn.proj", C, C)
                _ln(d, b + "norm2", C)
                _linear(d, b + "mlp.fc1", 4 * C, C)
                _linear(d, b + "mlp.fc2", C, 4 * C)
            _ln(d, g + "g2l_layer_norm", C)
            _conv(d, g + "embed_proj", C, 1, 1)          # present in checkpoints, unused at inference
        for i in range(len(inv)):
            double_conv_wobn(f"{p}convs.{i}", 2 * inv[i], inv[i], inv[i])


def patchfusion_spec(cfg):
    """cfg: the ``model.config`` mapping of a reference config file (plain or attribute dict)."""
    d = OrderedDict()
    cb, fb = cfg["coarse_branch"], cfg["fine_branch"]
    branch_spec(d, "coarse_branch.", cb)
    branch_spec(d, "fine_branch.", fb)
    C = branch_channels(fb)
    for i in range(6):
        if i == 5:
            _conv(d, f"fusion_conv_list.{i}", N_MIDAS_OUT, 2 * N_MIDAS_OUT, 3)
        else:
            _conv(d, f"fusion_conv_list.{i}", C, 2 * C, 3)
    guided_fusion_spec(d, "guided_fusion.", cfg["guided_fusion"])
    bins_head_spec(d, "", C, cb["n_bins"], cb["bin_embedding_dim"], cb["n_attractors"], cb.get("bin_centers_type", "softplus"))
    return d


def relative_position_index(win=G2L_WINDOW):
    """swin_layers.py:111-122 restated: index into the (2w-1)^2 bias table for every token pair."""
    ys, xs = torch.meshgrid(torch.arange(win), torch.arange(win), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    dy = ys[:, None] - ys[None, :] + win - 1
    dx = xs[:, None] - xs[None, :] + win - 1
    return dy * (2 * win - 1) + dx


def _seed_for(name, seed):
    import zlib
    return (zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF


def synthetic_state_dict(spec, seed=0, prefix_filter=None):
    """Deterministic, name-keyed synthetic weights (no network / checkpoint is available offline).

    Scaled so that activations stay O(1) through the whole net -- the reference's own random init
    (trunc_normal std=.02) makes the final depth nearly constant, which would
    make parity tests blind.  Every tensor depends only on (name, seed), so the container that
    generated tests/golden/ and the GPU box reproduce identical weights without shipping them.
    """
    out = OrderedDict()
    for name, e in spec.items():
        if prefix_filter is not None and not name.startswith(prefix_filter):
            continue
        g = torch.Generator().manual_seed(_seed_for(name, seed))
        shape = e.shape
        if e.kind == "w":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if "resize_layers.0" in name or "resize_layers.1" in name:
                fan_in = shape[0]             # ConvTranspose k==s: one tap per output pixel
            t = torch.randn(shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5
        elif e.kind == "b":
            t = torch.randn(shape, generator=g) * 0.1
            if name.endswith("conditional_log_binomial.mlp.2.bias"):
                # low temperature (t ~ 0.3..2 instead of ~25): makes the bin softmax selective so the
                # final depth actually depends on p and the bin centres (a flat softmax hides errors)
                t = t + torch.tensor([0.0, 0.0, -4.0, 1.0])
        elif e.kind == "ln_w":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif e.kind == "ln_b":
            t = 0.1 * torch.randn(shape, generator=g)
        elif e.kind == "gamma":
            t = 0.3 + 0.4 * torch.rand(shape, generator=g)
        elif e.kind == "pos":
            t = 0.2 * torch.randn(shape, generator=g)
        elif e.kind == "cls":
            t = 0.2 * torch.randn(shape, generator=g)
        elif e.kind == "relpos":
            t = 0.5 * torch.randn(shape, generator=g)
        elif e.kind == "bn_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif e.kind == "bn_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif e.kind == "zero_int":
            t = torch.zeros(shape, dtype=torch.int64)
        elif e.kind == "k_idx":
            t = torch.arange(shape[1]).view(shape)
        elif e.kind == "k_minus_1":
            n_bins = spec[name.replace("K_minus_1", "k_idx")].shape[1]
            t = torch.full(shape, float(n_bins - 1))
        elif e.kind == "relidx":
            t = relative_position_index()
        else:
            raise ValueError(e.kind)
        out[name] = t.to(e.dtype)
    return out
