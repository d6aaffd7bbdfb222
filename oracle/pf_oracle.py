"""TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product path (patchfusion_amd/).

CPU (torch fp32, device-agnostic) restatement of the reference's tiled-inference hot path --
``PatchFusion.forward(mode='infer')`` -- as pure functions over a flat ``state_dict``.
Every function cites the reference file:line (relative to /root/reference) it restates.

Pinned against the reference itself: tests/test_oracle_golden.py / tests/test_baseline_cpu.py run the reference's own
Python (oracle/ref_shim.py, only possible in the build container) on the same seeded weights and
inputs and requires agreement to 1e-5; tests/golden/*.npz hold outputs of the reference generated
by oracle/make_golden.py, so the check also runs where /root/reference is absent.
The two third-party leaves (torchvision roi_align, cv2.GaussianBlur) are PARITY UNPINNED
(oracle/third_party.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import third_party as tp

VIT_ARCH = {"vits": (384, 12, 6), "vitb": (768, 12, 12), "vitl": (1024, 24, 16)}
DPT_FEATURES = {"vits": 64, "vitb": 128, "vitl": 256}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def up(x, size):
    """F.interpolate(..., mode='bilinear', align_corners=True) -- used everywhere in the path."""
    return F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear", align_corners=True)


def conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def layer_norm(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


# ------------------------------------------------------------------------------------------
# DINOv2 ViT encoder
# ------------------------------------------------------------------------------------------
def vit_pos_embed(sd, p, th, tw):
    """vision_transformer.py:179-210 interpolate_pos_encoding (offset 0.1, bicubic, scale_factor)."""
    pos = sd[p + "pos_embed"].float()
    n = pos.shape[1] - 1
    g = int(math.sqrt(n))
    if th * tw == n and th == tw:
        return pos
    dim = pos.shape[-1]
    # reference passes (w, h) = x.shape[2:] i.e. w := image height, h := image width
    w0, h0 = th + 0.1, tw + 0.1
    grid = pos[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(float(w0) / g, float(h0) / g), mode="bicubic", antialias=False)
    assert grid.shape[-2] == th and grid.shape[-1] == tw
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([pos[:, :1], grid], dim=1)


def vit_attention(sd, b, x, heads):
    """dinov2/layers/attention.py:49-62 (exact softmax path; xFormers absent)."""
    B, N, C = x.shape
    qkv = linear(sd, b + "attn.qkv", x).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    x = (a @ v).transpose(1, 2).reshape(B, N, C)
    return linear(sd, b + "attn.proj", x)


def vit_block(sd, b, x, heads):
    """dinov2/layers/block.py:82-107 (eval branch), LayerScale layer_scale.py:27-28, Mlp mlp.py:35-41."""
    x = x + sd[b + "ls1.gamma"] * vit_attention(sd, b, layer_norm(sd, b + "norm1", x, 1e-6), heads)
    h = linear(sd, b + "mlp.fc1", layer_norm(sd, b + "norm2", x, 1e-6))
    h = linear(sd, b + "mlp.fc2", F.gelu(h))
    return x + sd[b + "ls2.gamma"] * h


def vit_forward(sd, p, x, enc, taps=None):
    """get_intermediate_layers(x, 4, return_class_token=True) -- vision_transformer.py:297-321:
    the LAST four blocks, each through the final norm, cls token dropped (dpt.py:100-105)."""
    D, depth, heads = VIT_ARCH[enc]
    B, _, H, W = x.shape
    th, tw = H // 14, W // 14
    tok = conv(sd, p + "patch_embed.proj", x, stride=14).flatten(2).transpose(1, 2)      # patch_embed.py:69-82
    tok = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), tok], dim=1)
    tok = tok + vit_pos_embed(sd, p, th, tw)
    if taps is not None:
        taps["vit_tokens_in"] = tok
    outs = []
    for i in range(depth):
        tok = vit_block(sd, f"{p}blocks.{i}.", tok, heads)
        if taps is not None and i in (0, depth - 1):
            taps[f"vit_block{i}"] = tok
        if i >= depth - 4:
            outs.append(layer_norm(sd, p + "norm", tok, 1e-6)[:, 1:])
    return outs, th, tw


# ------------------------------------------------------------------------------------------
# DPT head
# ------------------------------------------------------------------------------------------
def rcu(sd, name, x):
    """blocks.py:69-92 ResidualConvUnit with nn.ReLU(False), no bn."""
    out = conv(sd, name + ".conv1", F.relu(x), padding=1)
    out = conv(sd, name + ".conv2", F.relu(out), padding=1)
    return out + x


def fusion_block(sd, name, x, skip, size):
    """blocks.py:126-153 FeatureFusionBlock (align_corners=True)."""
    if skip is not None:
        x = x + rcu(sd, name + ".resConfUnit1", skip)
    x = rcu(sd, name + ".resConfUnit2", x)
    if size is None:
        size = (x.shape[-2] * 2, x.shape[-1] * 2)
    return conv(sd, name + ".out_conv", up(x, size))


def dpt_forward(sd, p, feats, th, tw, taps=None):
    """dpt.py:97-130 DPTHead.forward + dpt.py:146-156; returns rel_depth and the six hooked
    activations (depth_anything.py:299-321): out_conv, l4_rn, r4, r3, r2, r1."""
    B = feats[0].shape[0]
    maps = []
    for i, t in enumerate(feats):
        x = t.permute(0, 2, 1).reshape(B, t.shape[-1], th, tw)
        x = conv(sd, f"{p}projects.{i}", x)
        if i == 0:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.0.weight"], sd[p + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.1.weight"], sd[p + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = conv(sd, p + "resize_layers.3", x, stride=2, padding=1)
        maps.append(x)
    rn = [conv(sd, f"{p}scratch.layer{i + 1}_rn", maps[i], padding=1) for i in range(4)]
    r4 = fusion_block(sd, p + "scratch.refinenet4", rn[3], None, rn[2].shape[2:])
    r3 = fusion_block(sd, p + "scratch.refinenet3", r4, rn[2], rn[1].shape[2:])
    r2 = fusion_block(sd, p + "scratch.refinenet2", r3, rn[1], rn[0].shape[2:])
    r1 = fusion_block(sd, p + "scratch.refinenet1", r2, rn[0], None)
    out = conv(sd, p + "scratch.output_conv1", r1, padding=1)
    out = up(out, (th * 14, tw * 14))
    out_conv = F.relu(conv(sd, p + "scratch.output_conv2.0", out, padding=1))           # hook 'out_conv'
    rel = F.relu(conv(sd, p + "scratch.output_conv2.2", out_conv))
    rel = F.relu(up(rel, (th * 14, tw * 14)))                                           # dpt.py:154-156
    if taps is not None:
        taps.update(dpt_layer1_rn=rn[0], dpt_layer4_rn=rn[3])
    return rel.squeeze(1), dict(out_conv=out_conv, l4_rn=rn[3], r4=r4, r3=r3, r2=r2, r1=r1)


# ------------------------------------------------------------------------------------------
# Metric-bins head (ZoeDepth)
# ------------------------------------------------------------------------------------------
def mlp1x1(sd, name, x, act_out=None):
    """localbins_layers.py:71-117 / attractor.py:156-161: 1x1 conv, ReLU, 1x1 conv[, act]."""
    x = conv(sd, name + "._net.2", F.relu(conv(sd, name + "._net.0", x)))
    if act_out == "softplus":
        x = F.softplus(x)
    return x


def _attractor_delta(A, b_c, bcfg):
    """attractor.py:112-129 / :190-207: dist(A_i - c_j) reduced over the attractors.  NOTE both layers call ``dist(...)`` WITHOUT
    alpha/gamma, so the jit functions' defaults alpha=300, gamma=2 apply (config attractor_alpha=1000 / attractor_gamma are
    ignored); attractor_type 'exp' -> exp_attractor (:29-41), anything else -> inv_attractor (:44-57); kind 'mean' | 'sum'."""
    dx = A.unsqueeze(2) - b_c.unsqueeze(1)
    if bcfg.get("attractor_type", "inv") == "exp":
        d = torch.exp(-300.0 * (torch.abs(dx) ** 2)) * dx
    else:
        d = dx / (1 + 300.0 * dx.pow(2))
    return {"mean": torch.mean, "sum": torch.sum}[bcfg.get("attractor_kind", "mean")](d, dim=1)


def attractor_unnormed(sd, name, x, b_prev, prev_emb, bcfg):
    """attractor.py:164-208 AttractorLayerUnnormed.forward -> (b_new_centers, B_centers) (the same tensor)."""
    x = x + up(prev_emb, x.shape[-2:])
    A = mlp1x1(sd, name, x, "softplus")                                   # [B, n_attr, h, w]
    b_c = up(b_prev, A.shape[-2:])                                        # [B, n_bins, h, w]
    b_new = b_c + _attractor_delta(A, b_c, bcfg)
    return b_new, b_new


def attractor_normed(sd, name, x, b_prev, prev_emb, bcfg, min_depth, max_depth):
    """attractor.py:60-136 AttractorLayer.forward (bounded centres).  NOTE :105-106: the linear normalisation A / A.sum(dim=2)
    is computed and then OVERWRITTEN by A[:, :, 0] -- the attractor points are the even output channels of the MLP
    (ReLU + 1e-3), unnormalised.  Returns (b_new_centers normed, B_centers = clip(sort(scale(b_new))))."""
    x = x + up(prev_emb, x.shape[-2:])
    A = F.relu(mlp1x1(sd, name, x)) + 1e-3                                # [B, 2*n_attr, h, w]
    n, c, h, w = A.shape
    A = A.view(n, c // 2, 2, h, w)[:, :, 0]
    b_c = up(b_prev, (h, w))
    b_new = b_c + _attractor_delta(A, b_c, bcfg)
    Bc = (max_depth - min_depth) * b_new + min_depth
    Bc, _ = torch.sort(Bc, dim=1)
    return b_new, torch.clip(Bc, min_depth, max_depth)


def seed_bins_normed(sd, name, x, min_depth, max_depth):
    """localbins_layers.py:29-68 SeedBinRegressor.forward -> B_centers (bounded on (min_depth, max_depth))."""
    Bn = F.relu(mlp1x1(sd, name, x)) + 1e-3
    widths = (max_depth - min_depth) * (Bn / Bn.sum(dim=1, keepdim=True))
    widths = F.pad(widths, (0, 0, 0, 0, 1, 0), mode="constant", value=min_depth)
    edges = torch.cumsum(widths, dim=1)
    return 0.5 * (edges[:, :-1] + edges[:, 1:])


def log_binomial_depth(sd, p, last, emb, centers, min_temp, max_temp, n_bins=64, taps=None):
    """dist_layers.py:100-121 ConditionalLogBinomial + :51-69 LogBinomial + :29-33 log_binom, then
    zoedepth_v1.py:215-219 depth = sum_k prob_k * centers_k."""
    pt = conv(sd, p + "conditional_log_binomial.mlp.2",
              F.gelu(conv(sd, p + "conditional_log_binomial.mlp.0", torch.cat([last, emb], dim=1))))
    pt = F.softplus(pt)
    pp = pt[:, :2] + 1e-4
    prob = pp[:, 0] / (pp[:, 0] + pp[:, 1])
    tt = pt[:, 2:] + 1e-4
    t = (tt[:, 0] / (tt[:, 0] + tt[:, 1])).unsqueeze(1)
    t = (max_temp - min_temp) * t + min_temp
    x = prob.unsqueeze(1)
    one_minus = torch.clamp(1 - x, 1e-4, 1)
    x = torch.clamp(x, 1e-4, 1)
    k = torch.arange(n_bins, device=x.device, dtype=torch.float32).view(1, -1, 1, 1)
    K1 = torch.full((1, 1, 1, 1), float(n_bins - 1), device=x.device)
    eps = 1e-7
    n_, k_ = K1 + eps, k + eps
    logc = n_ * torch.log(n_) - k_ * torch.log(k_) - (n_ - k_) * torch.log(n_ - k_ + eps)
    y = logc + k * torch.log(x) + (n_bins - 1 - k) * torch.log(one_minus)
    probs = torch.softmax(y / t, dim=1)
    centers = up(centers, probs.shape[-2:])
    if taps is not None:
        taps.update(clb_p=prob, clb_t=t)
    return torch.sum(probs * centers, dim=1, keepdim=True)


def bins_head(sd, p, x0, x_blocks, last, rel_cond, bcfg, taps=None, min_depth=None, max_depth=None):
    """zoedepth_v1.py:173-219 == patchfusion.py:297-340 (same math, different weights / inputs).  bin_centers_type
    (zoedepth_v1.py:90-104 == patchfusion.py:132-146): 'softplus' (all shipped configs) | 'normed' | 'hybrid1' | 'hybrid2' picks the
    seed regressor and the attractor layer; min_depth / max_depth only matter for the bounded variants (default: the branch
    config's own)."""
    kind = bcfg.get("bin_centers_type", "softplus")
    if kind not in ("normed", "softplus", "hybrid1", "hybrid2"):
        raise ValueError("bin_centers_type should be one of 'normed', 'softplus', 'hybrid1', 'hybrid2'")
    lo = float(bcfg.get("min_depth", 1e-3) if min_depth is None else min_depth)
    hi = float(bcfg.get("max_depth", 10) if max_depth is None else max_depth)
    if kind in ("normed", "hybrid1"):
        b_prev = seed_bins_normed(sd, p + "seed_bin_regressor", x0, lo, hi)
    else:
        b_prev = mlp1x1(sd, p + "seed_bin_regressor", x0, "softplus")
    if kind in ("normed", "hybrid2"):                                      # zoedepth_v1.py:178-182
        b_prev = (b_prev - lo) / (hi - lo)
    prev_emb = mlp1x1(sd, p + "seed_projector", x0)
    emb = centers = None
    for i, xb in enumerate(x_blocks):
        emb = mlp1x1(sd, f"{p}projectors.{i}", xb)
        if kind in ("normed", "hybrid2"):
            b_prev, centers = attractor_normed(sd, f"{p}attractors.{i}", emb, b_prev, prev_emb, bcfg, lo, hi)
        else:
            b_prev, centers = attractor_unnormed(sd, f"{p}attractors.{i}", emb, b_prev, prev_emb, bcfg)
        prev_emb = emb
        if taps is not None:
            taps[f"bins_centers{i}"] = b_prev
    rel_cond = up(rel_cond, last.shape[2:])
    last = torch.cat([last, rel_cond], dim=1)
    emb = up(emb, last.shape[-2:])
    return log_binomial_depth(sd, p, last, emb, centers, bcfg["min_temp"], bcfg["max_temp"], bcfg["n_bins"], taps)


def branch_forward(sd, p, x, bcfg, taps=None):
    """ZoeDepth.forward(x, return_final_centers=True) for a Depth-Anything core:
    zoedepth_v1.py:125-233 + depth_anything.py:262-278 (normalise only; do_resize=False).
    Returns metric_depth [B,1,H,W] and the six temp_features low->high
    (x_d0, r4, r3, r2, r1, out_conv) -- patchfusion.py:189-206."""
    enc = bcfg["midas_model_type"]
    mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device).view(1, 3, 1, 1)
    xn = (x - mean) / std
    feats, th, tw = vit_forward(sd, p + "core.core.pretrained.", xn, enc, taps)
    if taps is not None:
        for i, f in enumerate(feats):
            taps[f"vit_out{i}"] = f
    rel, hk = dpt_forward(sd, p + "core.core.depth_head.", feats, th, tw, taps)
    x_d0 = conv(sd, p + "conv2", hk["l4_rn"])
    x_blocks = [hk["r4"], hk["r3"], hk["r2"], hk["r1"]]
    depth = bins_head(sd, p, x_d0, x_blocks, hk["out_conv"], rel.unsqueeze(1), bcfg, taps)
    if taps is not None:
        taps["rel_depth"] = rel
    return depth, [x_d0] + x_blocks + [hk["out_conv"]]


def branch_forward_external(sd, p, x, bcfg, provider, taps=None):
    """ZoeDepth.forward for a MiDaS-core branch (type 'ZoeDepth', BASELINE configs[4]) with the relative-depth core supplied
    from OUTSIDE -- the reference's own injection point `hack_feature=(rel_depth, out)`, zoedepth_v1.py:160-166:
    provider(x) -> (rel_depth [B,H,W], [btlnck, x_block0..3, outconv_activation]).  PARITY UNPINNED for the core itself: it is
    the un-vendored torch.hub repo AyaanShah2204/MiDaS (midas.py:340); everything after it (conv2, the metric-bins head) is
    the same arithmetic as the Depth-Anything branch: zoedepth_v1.py:170-219."""
    rel, out = provider(x)
    btl, x_blocks, last = out[0], list(out[1:5]), out[5]
    x_d0 = conv(sd, p + "conv2", btl)
    depth = bins_head(sd, p, x_d0, x_blocks, last, rel.unsqueeze(1), bcfg, taps)
    return depth, [x_d0] + x_blocks + [last]


def any_branch_forward(sd, p, x, bcfg, provider=None, taps=None):
    if bcfg["type"] == "ZoeDepth":
        return branch_forward_external(sd, p, x, bcfg, provider, taps)
    return branch_forward(sd, p, x, bcfg, taps)


# ------------------------------------------------------------------------------------------
# G2L (Swin) + guided fusion
# ------------------------------------------------------------------------------------------
def swin_shift_mask(Hp, Wp, win, shift, device):
    """swin_layers.py:327-345: region ids then 0 / -100 mask per window pair."""
    img = torch.zeros((Hp, Wp), device=device)
    cnt = 0
    for hs in (slice(0, -win), slice(-win, -shift), slice(-shift, None)):
        for ws in (slice(0, -win), slice(-win, -shift), slice(-shift, None)):
            img[hs, ws] = cnt
            cnt += 1
    mw = img.view(Hp // win, win, Wp // win, win).permute(0, 2, 1, 3).reshape(-1, win * win)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(m != 0, torch.full_like(m, -100.0), torch.zeros_like(m))


def swin_block(sd, b, x, H, W, heads, win, shift, mask):
    """swin_layers.py:218-268 + WindowAttention :133-164.  Padding (zeros) is applied AFTER norm1
    and padded tokens attend unmasked in non-shifted blocks."""
    B, L, C = x.shape
    shortcut = x
    x = layer_norm(sd, b + "norm1", x, 1e-5).view(B, H, W, C)
    pr, pb = (win - W % win) % win, (win - H % win) % win
    x = F.pad(x, (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.view(B, Hp // win, win, Wp // win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, win * win, C)
    nW, N = xw.shape[0], win * win
    hd = C // heads
    qkv = linear(sd, b + "attn.qkv", xw).reshape(nW, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[b + "attn.relative_position_bias_table"][sd[b + "attn.relative_position_index"].view(-1)]
    attn = attn + bias.view(N, N, heads).permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        attn = attn.view(B, nW // B, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(dim=-1)
    xw = linear(sd, b + "attn.proj", (attn @ v).transpose(1, 2).reshape(nW, N, C))
    x = xw.view(B, Hp // win, Wp // win, win, win, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].reshape(B, H * W, C)
    x = shortcut + x
    h = linear(sd, b + "mlp.fc2", F.gelu(linear(sd, b + "mlp.fc1", layer_norm(sd, b + "norm2", x, 1e-5))))
    return x + h


def g2l_forward(sd, g, feat, heads, depth, win=12):
    """swin_layers.py:410-432 G2LFusion.forward(x, None) (+ G2LBasicLayer :325-355)."""
    B, C, H, W = feat.shape
    x = feat.flatten(2).transpose(1, 2) + sd[g + "absolute_pos_embed"]
    Hp, Wp = int(math.ceil(H / win)) * win, int(math.ceil(W / win)) * win
    mask = swin_shift_mask(Hp, Wp, win, win // 2, feat.device)
    for j in range(depth):
        x = swin_block(sd, f"{g}g2l_layer.blocks.{j}.", x, H, W, heads, win, 0 if j % 2 == 0 else win // 2, mask)
    x = layer_norm(sd, g + "g2l_layer_norm", x, 1e-5)
    return x.view(B, H, W, C).permute(0, 3, 1, 2).contiguous()


def double_conv_bn(sd, name, x):
    """guided_fusion_model.py:52-69 DoubleConv in eval mode (running stats, eps 1e-5)."""
    for ci, bi in ((0, 1), (3, 4)):
        x = conv(sd, f"{name}.double_conv.{ci}", x, padding=1)
        n = f"{name}.double_conv.{bi}"
        x = F.batch_norm(x, sd[n + ".running_mean"], sd[n + ".running_var"], sd[n + ".weight"], sd[n + ".bias"],
                         training=False, eps=1e-5)
        x = F.relu(x)
    return x


def double_conv_wobn(sd, name, x):
    """guided_fusion_model.py:34-50."""
    x = F.relu(conv(sd, name + ".double_conv.0", x, padding=1))
    return F.relu(conv(sd, name + ".double_conv.2", x, padding=1))


G2L_DEPTH_INV = [4, 4, 3, 3, 2, 2]
G2L_HEADS_INV = [32, 32, 16, 16, 8, 8]


def g2l_all(sd, coarse_feats, p="guided_fusion."):
    """The six G2L outputs depend only on the whole-image coarse pyramid
    (guided_fusion_model.py:188,200-201), i.e. are identical for every patch batch of an image."""
    return [g2l_forward(sd, f"{p}g2l_list.{i}.", f, G2L_HEADS_INV[i], G2L_DEPTH_INV[i]) for i, f in enumerate(coarse_feats)]


def guided_fusion_forward(sd, inp, guide_cat, bbox, coarse_feats, process_h, p="guided_fusion.", g2l_cache=None, taps=None):
    """guided_fusion_model.py:163-207. Returns the six fused maps low->high (the reference
    returns output[::-1] and the caller reverses again, patchfusion.py:273-282)."""
    enc = [double_conv_bn(sd, p + "inc", inp)]
    for i in range(5):
        enc.append(double_conv_bn(sd, f"{p}down_conv_list.{i}.maxpool_conv.1", F.max_pool2d(enc[-1], 2)))
    enc = enc[::-1]
    g2l = g2l_cache if g2l_cache is not None else g2l_all(sd, coarse_feats, p)
    out, temp = [], None
    for i, (fe, fc) in enumerate(zip(enc, coarse_feats)):
        h, w = fc.shape[-2:]
        if fe.shape[-2:] != fc.shape[-2:]:
            fe = up(fe, (h, w))
        if i > 0:
            x1 = up(torch.cat([temp, guide_cat[i - 1]], dim=1), (h, w))              # Upv1 :96-100
            fe = double_conv_wobn(sd, f"{p}up_conv_list.{i - 1}.conv", torch.cat([fe, x1], dim=1))
        fg = tp.roi_align(g2l[i], bbox, (h, w), h / process_h, aligned=True)
        temp = double_conv_wobn(sd, f"{p}convs.{i}", torch.cat([fe, fg], dim=1))
        if taps is not None:
            taps[f"gf_out{i}"] = temp
        out.append(temp)
    return out


def coarse_rois(coarse_depth, coarse_feats, bboxs_feat, process_h):
    """patchfusion.py:240-257 coarse_postprocess_test: crop-resize the patch's area of every coarse
    level (and the coarse depth) back to the level's full grid.  ``feat.repeat(P,...)`` + batch index
    k is equivalent to batch index 0 on the un-repeated tensor."""
    rois = bboxs_feat.clone()
    rois[:, 0] = 0
    feats = [tp.roi_align(f, rois, f.shape[-2:], f.shape[-2] / process_h, aligned=True) for f in coarse_feats]
    depth = tp.roi_align(coarse_depth, rois, coarse_depth.shape[-2:], coarse_depth.shape[-2] / process_h, aligned=True)
    return depth, feats


def fusion_forward(sd, cfg, fine_depth, crops, coarse_feats, fine_feats, bbox, coarse_depth_roi, coarse_feats_roi,
                   g2l_cache=None, taps=None):
    """patchfusion.py:259-340."""
    cat = [conv(sd, f"fusion_conv_list.{i}", torch.cat([c, f], dim=1), padding=1)
           for i, (c, f) in enumerate(zip(coarse_feats_roi, fine_feats))]
    inp = torch.cat([coarse_depth_roi, fine_depth, crops], dim=1)
    fused = guided_fusion_forward(sd, inp, cat, bbox, coarse_feats, cfg["patch_process_shape"][0],
                                  g2l_cache=g2l_cache, taps=taps)
    last = fused[-1]
    rel_cond = torch.zeros((last.shape[0], 1) + tuple(last.shape[-2:]), device=last.device)
    return bins_head(sd, "", fused[0], fused[1:-1], last, rel_cond, cfg["coarse_branch"], None,
                     min_depth=cfg["min_depth"], max_depth=cfg["max_depth"])       # patchfusion.py:152-163: config.min/max_depth


def silog_loss(pred, target, min_depth, max_depth, beta=0.15):
    """SILogLoss.forward -- estimator/models/losses.py:23-46 (no additional_mask)."""
    if pred.shape[-2:] != target.shape[-2:]:
        pred = up(pred, target.shape[-2:])
    mask = torch.logical_and(target > min_depth, target < max_depth)
    if torch.sum(mask) <= 1:
        return pred * 0.0
    g = torch.log(pred[mask] + 1e-7) - torch.log(target[mask] + 1e-7)
    return 10 * torch.sqrt(torch.var(g) + beta * torch.pow(torch.mean(g), 2))


# ------------------------------------------------------------------------------------------
# Tiling / stitching
# ------------------------------------------------------------------------------------------
def generatemask(size):
    """estimator/models/utils.py:38-47."""
    mask = np.zeros(size, dtype=np.float32)
    sigma = int(size[0] / 16)
    k = int(2 * np.ceil(2 * int(size[0] / 16)) + 1)
    mask[int(0.1 * size[0]):size[0] - int(0.1 * size[0]), int(0.1 * size[1]):size[1] - int(0.1 * size[1])] = 1
    mask = tp.gaussian_blur(mask, (k, k), sigma)
    mask = (mask - mask.min()) / (mask.max() - mask.min())
    return mask.astype(np.float32)


def prepare_tile_cfg(process_shape, image_raw_shape, split):
    """baseline_pretrain.py:91-119."""
    assert image_raw_shape[0] % (2 * split[0]) == 0 and image_raw_shape[1] % (2 * split[1]) == 0
    return dict(patch_split_num=tuple(split),
                patch_reensemble_shape=(process_shape[0] * split[0], process_shape[1] * split[1]),
                patch_raw_shape=(image_raw_shape[0] // split[0], image_raw_shape[1] // split[1]),
                image_raw_shape=tuple(image_raw_shape))


class RunningAverageMap:
    """estimator/models/utils.py:21-36."""

    def __init__(self, average_map, count_map):
        self.count_map = count_map
        self.average_map = average_map / count_map

    def update(self, pred_map, ct_map):
        self.average_map = (pred_map + self.count_map * self.average_map) / (self.count_map + ct_map)
        self.count_map = self.count_map + ct_map

    def resize(self, resolution):
        a = self.average_map[None, None]
        c = self.count_map[None, None]
        self.average_map = F.interpolate(a, size=tuple(resolution)).squeeze()          # nearest
        self.count_map = up(c, resolution).squeeze()


class Oracle:
    """Stateful wrapper reproducing PatchFusion.forward(mode='infer') -- patchfusion.py:401-453."""

    def __init__(self, cfg, sd, hoist_g2l=True, core_providers=None):
        """core_providers = (coarse, fine) feature providers, only for type 'ZoeDepth' branches (branch_forward_external)"""
        self.cfg, self.sd = cfg, sd
        self.providers = core_providers or (None, None)
        self.ps = tuple(cfg["patch_process_shape"])
        self.hoist_g2l = hoist_g2l      # exact algebraic saving; False follows the reference schedule
        self.taps = None

    def resizer(self, x):
        """depth_anything/transform.py:127-129 with keep_aspect_ratio=False: plain resize to the
        process shape (a multiple of 14)."""
        return up(x, self.ps)

    def _bbox_feat(self, bboxs, tile_cfg):
        H, W = tile_cfg["image_raw_shape"]
        fac = torch.tensor([1 / W * self.ps[1], 1 / H * self.ps[0], 1 / W * self.ps[1], 1 / H * self.ps[0]],
                           device=bboxs.device).unsqueeze(0)
        bf = bboxs * fac
        inds = torch.arange(bboxs.shape[0], device=bboxs.device).unsqueeze(-1)
        return torch.cat([inds, bf], dim=-1)

    def _predict(self, crops, bboxs, tile_cfg, process_num):
        """Batched coarse-roi + fine + fusion (baseline_pretrain.py:275-307, patchfusion.py:343-356)."""
        bf = self._bbox_feat(bboxs, tile_cfg)
        d_roi, f_roi = coarse_rois(self.coarse_depth, self.coarse_feats, bf, self.ps[0])
        preds = []
        for s in range(0, crops.shape[0], process_num):
            sl = slice(s, s + process_num)
            fine_depth, fine_feats = any_branch_forward(self.sd, "fine_branch.", crops[sl], self.cfg["fine_branch"], self.providers[1])
            bb = bf[sl].clone()
            bb[:, 0] = 0
            preds.append(fusion_forward(self.sd, self.cfg, fine_depth, crops[sl], self.coarse_feats, fine_feats, bb,
                                        d_roi[sl], [f[sl] for f in f_roi], g2l_cache=self.g2l, taps=self.taps))
        return torch.cat(preds, dim=0)

    def regular_tile(self, offset, offset_process, image_hr, init_flag, blur_mask, avg, tile_cfg, process_num):
        """baseline_pretrain.py:222-331."""
        h_raw, w_raw = tile_cfg["patch_raw_shape"]
        H, W = tile_cfg["image_raw_shape"]
        hs = [h_raw * i + offset[0] for i in range((H - offset[0]) // h_raw)]
        ws = [w_raw * i + offset[1] for i in range((W - offset[1]) // w_raw)]
        RH, RW = tile_cfg["patch_reensemble_shape"]
        hp = [self.ps[0] * i + offset_process[0] for i in range((RH - offset_process[0]) // self.ps[0])]
        wp = [self.ps[1] * i + offset_process[1] for i in range((RW - offset_process[1]) // self.ps[1])]
        crops, bboxs = [], []
        for h in hs:
            for w in ws:
                crops.append(self.resizer(image_hr[None, :, h:h + h_raw, w:w + w_raw])[0])
                bboxs.append([w, h, w + w_raw, h + h_raw])
        crops = torch.stack(crops)
        bboxs = torch.tensor(bboxs, device=image_hr.device).int()
        preds = self._predict(crops, bboxs, tile_cfg, process_num)
        count = torch.zeros((RH, RW), device=image_hr.device)
        pred = torch.zeros((RH, RW), device=image_hr.device)
        idx = 0
        for h in hp:
            for w in wp:
                d = preds[idx, 0]
                if not init_flag:
                    count = torch.zeros((RH, RW), device=image_hr.device)
                    pred = torch.zeros((RH, RW), device=image_hr.device)
                count[h:h + self.ps[0], w:w + self.ps[1]] = blur_mask
                pred[h:h + self.ps[0], w:w + self.ps[1]] = d * blur_mask
                if not init_flag:
                    avg.update(pred, count)
                idx += 1
        if init_flag:
            avg = RunningAverageMap(pred, count)
        return avg

    def random_tile(self, image_hr, blur_mask, avg, tile_cfg, process_num):
        """baseline_pretrain.py:143-218: process_num random h_starts, ONE shared w_start (:155-156);
        prediction nearest-resized to the raw patch size (:203)."""
        h_raw, w_raw = tile_cfg["patch_raw_shape"]
        H, W = tile_cfg["image_raw_shape"]
        hs = [random.randint(0, H - h_raw - 1) for _ in range(process_num)]
        ws = [random.randint(0, W - w_raw - 1)]
        crops, bboxs = [], []
        for h in hs:
            for w in ws:
                crops.append(self.resizer(image_hr[None, :, h:h + h_raw, w:w + w_raw])[0])
                bboxs.append([w, h, w + w_raw, h + h_raw])
        crops = torch.stack(crops)
        bboxs = torch.tensor(bboxs, device=image_hr.device).int()
        preds = self._predict(crops, bboxs, tile_cfg, process_num)
        preds = F.interpolate(preds, (h_raw, w_raw))
        idx = 0
        for h in hs:
            for w in ws:
                count = torch.zeros((H, W), device=image_hr.device)
                pred = torch.zeros((H, W), device=image_hr.device)
                count[h:h + h_raw, w:w + w_raw] = blur_mask
                pred[h:h + h_raw, w:w + w_raw] = preds[idx, 0] * blur_mask
                avg.update(pred, count)
                idx += 1
        return avg

    @torch.no_grad()
    def train_forward(self, image_lr, crops_image_hr, crop_depths, bboxs):
        """PatchFusion.forward(mode='train') -- patchfusion.py:372-399: coarse branch on the batch of low-resolution images, fine
        branch on one crop per image, coarse_postprocess_train (:227-237: roi_align with batch index i), fusion_forward,
        SILogLoss (losses.py:15-62).  Forward value only.  Returns (sig_loss, depth_prediction [B,1,h,w])."""
        cfg, sd = self.cfg, self.sd
        H, W = cfg["image_raw_shape"]
        fac = torch.tensor([1 / W * self.ps[1], 1 / H * self.ps[0], 1 / W * self.ps[1], 1 / H * self.ps[0]], device=bboxs.device).unsqueeze(0)
        bf = torch.cat((torch.arange(bboxs.shape[0], device=bboxs.device).unsqueeze(-1), bboxs * fac), dim=-1)
        cdepth, cfeats = any_branch_forward(sd, "coarse_branch.", image_lr, cfg["coarse_branch"], self.providers[0])
        fdepth, ffeats = any_branch_forward(sd, "fine_branch.", crops_image_hr, cfg["fine_branch"], self.providers[1])
        feats_roi = [tp.roi_align(f, bf, f.shape[-2:], f.shape[-2] / self.ps[0], aligned=True) for f in cfeats]
        depth_roi = tp.roi_align(cdepth, bf, cdepth.shape[-2:], cdepth.shape[-2] / self.ps[0], aligned=True)
        pred = fusion_forward(sd, cfg, fdepth, crops_image_hr, cfeats, ffeats, bf, depth_roi, feats_roi, g2l_cache=None)
        return silog_loss(pred, crop_depths, cfg["min_depth"], cfg["max_depth"]), pred

    @torch.no_grad()
    def infer(self, image_lr, image_hr, cai_mode="m1", process_num=4, tile_cfg=None, taps=None):
        cfg = self.cfg
        self.taps = taps
        if tile_cfg is None:
            tile_cfg = dict(image_raw_shape=cfg["image_raw_shape"], patch_split_num=cfg["patch_split_num"])
        tile_cfg = prepare_tile_cfg(self.ps, tile_cfg["image_raw_shape"], tile_cfg["patch_split_num"])
        assert image_hr.shape[0] == 1
        self.coarse_depth, self.coarse_feats = any_branch_forward(self.sd, "coarse_branch.", image_lr, cfg["coarse_branch"],
                                                                  self.providers[0], taps)
        if taps is not None:
            taps["coarse_depth"] = self.coarse_depth
            for i, f in enumerate(self.coarse_feats):
                taps[f"coarse_feat{i}"] = f
        self.g2l = g2l_all(self.sd, self.coarse_feats) if self.hoist_g2l else None
        if taps is not None and self.g2l is not None:
            for i, f in enumerate(self.g2l):
                taps[f"g2l{i}"] = f
        dev = image_hr.device
        blur = torch.tensor(generatemask(self.ps) + 1e-3, device=dev)
        img = image_hr[0]
        avg = self.regular_tile([0, 0], [0, 0], img, True, blur, None, tile_cfg, process_num)
        if cai_mode == "m2" or cai_mode[0] == "r":
            hr, wr = tile_cfg["patch_raw_shape"]
            for off, offp in (([0, wr // 2], [0, self.ps[1] // 2]), ([hr // 2, 0], [self.ps[0] // 2, 0]),
                              ([hr // 2, wr // 2], [self.ps[0] // 2, self.ps[1] // 2])):
                avg = self.regular_tile(off, offp, img, False, blur, avg, tile_cfg, process_num)
        if cai_mode[0] == "r":
            blur = torch.tensor(generatemask(tile_cfg["patch_raw_shape"]) + 1e-3, device=dev)
            avg.resize(tile_cfg["image_raw_shape"])
            for _ in range(int(cai_mode[1:]) // process_num):
                avg = self.random_tile(img, blur, avg, tile_cfg, process_num)
        return avg.average_map[None, None]


class BaselineOracle(Oracle):
    """BaselinePretrain.forward(mode='infer') -- baseline_pretrain.py:365-420: target='coarse' = the branch on
    image_lr; target='fine' = tiles -> fine branch -> stitch (no coarse pass, no fusion).  NOTE r<N> makes N
    random_tile calls here (:404-408), not N // process_num as in PatchFusion."""

    def __init__(self, cfg_branch, process_shape, image_raw_shape, split, sd, target, core_provider=None):
        self.bcfg, self.ps, self.sd, self.target = cfg_branch, tuple(process_shape), sd, target
        self.raw, self.split = tuple(image_raw_shape), tuple(split)
        self.provider = core_provider                      # type 'ZoeDepth': the external relative-depth core (hack_feature hook)
        self.taps = None

    def _branch(self, x):
        return any_branch_forward(self.sd, self.target + "_branch.", x, self.bcfg, self.provider)[0]

    def _predict(self, crops, bboxs, tile_cfg, process_num):
        preds = [self._branch(crops[s:s + process_num]) for s in range(0, crops.shape[0], process_num)]
        return torch.cat(preds, dim=0)

    @torch.no_grad()
    def train_forward(self, x, depth_gt, min_depth=1e-3, max_depth=80):
        """BaselinePretrain.forward(mode='train') -- baseline_pretrain.py:347-363: the branch on the batch (image_lr for target
        'coarse', crops_image_hr for 'fine') and SILogLoss against depth_gt / crop_depths.  Returns (loss, depth [B,1,h,w])."""
        pred = self._branch(x)
        return silog_loss(pred, depth_gt, min_depth, max_depth), pred

    @torch.no_grad()
    def infer(self, image_lr, image_hr, cai_mode="m1", process_num=4, tile_cfg=None):
        if self.target == "coarse":
            return self._branch(image_lr)
        tile_cfg = prepare_tile_cfg(self.ps, self.raw, self.split) if tile_cfg is None else \
            prepare_tile_cfg(self.ps, tile_cfg["image_raw_shape"], tile_cfg["patch_split_num"])
        dev = image_hr.device
        blur = torch.tensor(generatemask(self.ps) + 1e-3, device=dev)
        img = image_hr[0]
        avg = self.regular_tile([0, 0], [0, 0], img, True, blur, None, tile_cfg, process_num)
        if cai_mode == "m2" or cai_mode[0] == "r":
            hr, wr = tile_cfg["patch_raw_shape"]
            for off, offp in (([0, wr // 2], [0, self.ps[1] // 2]), ([hr // 2, 0], [self.ps[0] // 2, 0]),
                              ([hr // 2, wr // 2], [self.ps[0] // 2, self.ps[1] // 2])):
                avg = self.regular_tile(off, offp, img, False, blur, avg, tile_cfg, process_num)
        if cai_mode[0] == "r":
            blur = torch.tensor(generatemask(tile_cfg["patch_raw_shape"]) + 1e-3, device=dev)
            avg.resize(tile_cfg["image_raw_shape"])
            for _ in range(int(cai_mode[1:])):
                avg = self.random_tile(img, blur, avg, tile_cfg, process_num)
        return avg.average_map[None, None]
