"""Network executors of the PatchFusion hot path on top of the HIP op set.

Organised around fused stages with explicit NHWC buffers, not nn.Modules: every method below is a
straight-line sequence of kernel launches (``ops.*`` -> C ABI -> hand-written HIP).  ``ops`` is the
op back-end object; the product always passes :data:`patchfusion_amd.hip_ops.ops` (tests substitute
a torch reference back-end from tests/fake_ops.py to validate the wiring and weight packing on CPU,
and to serve as the per-op reference for the HIP kernels on the GPU).

Reference functions restated here (paths under the reference repo):
  BranchNet      external/zoedepth/models/zoedepth/zoedepth_v1.py:125-233 (ZoeDepth.forward),
                 base_models/depth_anything.py:262-278, depth_anything/dpt.py:97-156, blocks.py:69-153,
                 torchhub/.../vision_transformer.py:212-231,297-321, dinov2/layers/{block,attention,mlp}.py
  G2LNet         estimator/models/blocks/swin_layers.py:218-268,325-355,410-432
  FusionNet      estimator/models/patchfusion.py:240-340, blocks/guided_fusion_model.py:163-207
  BinsHead       zoedepth_v1.py:173-219 == patchfusion.py:297-340; layers/{localbins,attractor,dist}_layers.py
"""
import math

import torch

from . import packing as pk
from .config import pyramid_sizes
from .spec import DPT_ARCH, VIT_ARCH, branch_channels

F32 = torch.float32


def linear_split3_enabled():
    """float32 mode: run the ViT block linears as split-precision GEMMs (csrc/gemm_split3.hip)?  PF_LINEAR_SPLIT3=0 / 1; default on"""
    import os
    return os.environ.get("PF_LINEAR_SPLIT3", "1") != "0"


def attention_split3_enabled():
    """float32 mode with split-precision linears: also run the ViT attention in split precision (pf_vit_attention_split3)?  PF_ATTN_SPLIT3=0 / 1"""
    import os
    return os.environ.get("PF_ATTN_SPLIT3", "1") != "0"


def bins_tail_enabled():
    """float32 mode: the fused metric-bins tail kernel (pf_bins_tail)?  PF_BINS_TAIL=0 keeps the four separate launches"""
    import os
    return os.environ.get("PF_BINS_TAIL", "1") != "0"


def _g(sd, name, device):
    return sd[name].detach().to(device=device, dtype=F32).contiguous()


class BinsHead:
    """seed regressor/projector, 4 x (projector, attractor), conditional log-binomial, expectation."""

    def __init__(self, sd, prefix, C, bcfg, dtype, device, with_rel, min_depth=None, max_depth=None):
        self.dtype, self.device = dtype, device
        # bin_centers_type (zoedepth_v1.py:90-104 == patchfusion.py:132-146): bounded seed regressor for 'normed' / 'hybrid1', bounded
        # attractor layer for 'normed' / 'hybrid2'; min / max depth only matter for those (branch head: the branch config's own,
        # fusion head: the top-level config's, patchfusion.py:152-163)
        kind = bcfg.get("bin_centers_type", "softplus")
        if kind not in ("normed", "softplus", "hybrid1", "hybrid2"):
            raise ValueError("bin_centers_type should be one of 'normed', 'softplus', 'hybrid1', 'hybrid2'")
        self.seed_bounded, self.attr_bounded = kind in ("normed", "hybrid1"), kind in ("normed", "hybrid2")
        self.lo = float(bcfg.get("min_depth", 1e-3) if min_depth is None else min_depth)
        self.hi = float(bcfg.get("max_depth", 10) if max_depth is None else max_depth)
        # defaults of ZoeDepth.build for a branch head (zoedepth_v1.py:41: attractor_type 'exp', attractor_kind 'sum'); the fusion head
        # reads both straight from the config and fails without them (patchfusion.py:162)
        if not with_rel and not ("attractor_type" in bcfg and "attractor_kind" in bcfg):
            raise AttributeError("coarse_branch config needs attractor_type and attractor_kind for the fusion head (patchfusion.py:162)")
        self.attr_type, self.attr_kind = bcfg.get("attractor_type", "exp"), bcfg.get("attractor_kind", "sum")
        if self.attr_kind not in ("mean", "sum"):
            raise KeyError(self.attr_kind)                          # attractor.py:118 indexes a dict with it
        self.n_attr = list(bcfg["n_attractors"])
        self.n_bins = int(bcfg["n_bins"])
        self.emb = int(bcfg["bin_embedding_dim"])
        self.min_temp, self.max_temp = float(bcfg["min_temp"]), float(bcfg["max_temp"])
        self.with_rel = with_rel

        def pc(name, **kw):
            return pk.pack_conv(sd[prefix + name + ".weight"], sd[prefix + name + ".bias"], dtype=dtype, **kw).to(device)

        self.sbr0, self.sbr2 = pc("seed_bin_regressor._net.0"), pc("seed_bin_regressor._net.2")
        self.sp0, self.sp2 = pc("seed_projector._net.0"), pc("seed_projector._net.2")
        self.proj = [(pc(f"projectors.{i}._net.0"), pc(f"projectors.{i}._net.2")) for i in range(4)]
        self.attr = [(pc(f"attractors.{i}._net.0"), pc(f"attractors.{i}._net.2")) for i in range(4)]
        # CLB input buffer layout: [last 0..31 | emb 32..159 | rel 160 (+7 pad)]; the reference order is
        # cat([last(32), rel(1)], emb(128)) (zoedepth_v1.py:207-213, dist_layers.py:110)
        if with_rel:
            cmap, ctot = [(0, 32, 0), (32, 1, 32 + self.emb), (33, self.emb, 32)], 32 + self.emb + 8
        else:  # fusion head: rel_cond is all zeros (patchfusion.py:300) -> its column drops out exactly
            cmap, ctot = [(0, 32, 0), (33, self.emb, 32)], 32 + self.emb
        self.clb_channels = ctot
        self.mlp0 = pc("conditional_log_binomial.mlp.0", cin_map=cmap, cin_total=ctot)
        self.mlp2 = pc("conditional_log_binomial.mlp.2")
        # float32: resize-into-CLB + mlp.0 + mlp.2 + log-binomial as ONE kernel (pf_bins_tail) when the head has the shipped shape
        self.tail = None
        if dtype == F32 and self.n_bins == 64 and bins_tail_enabled():
            self.tail = pk.bins_tail_weights(self.mlp0, self.mlp2, self.emb)
            if self.tail is not None:
                self.tail = self.tail.to(device)

    def new_clb_buffer(self, ops, B, H, W):
        return ops.empty((B, H, W, self.clb_channels), self.dtype, self.device)

    def run(self, ops, x0, x_blocks, clb, taps=None):
        """x0 [B,h0,w0,C]; x_blocks 4 maps low->high; clb: buffer whose [..., :32] already holds `last`
        (and [..., 160:168] the relative depth when with_rel).  Returns depth f32 [B,H,W]."""
        dt, dev = self.dtype, self.device
        B, h0, w0, _ = x0.shape
        t = ops.empty((B, h0, w0, self.sbr0.cout), dt, dev)
        ops.conv(x0, self.sbr0, t, act="relu")
        b_prev = ops.empty((B, h0, w0, self.n_bins), F32, dev)
        ops.conv(t, self.sbr2, b_prev, act="relu" if self.seed_bounded else "softplus")
        if self.seed_bounded or self.attr_bounded:                  # localbins_layers.py:52-68, zoedepth_v1.py:178-182
            b_prev = ops.seed_bin_centers(b_prev, ops.empty((B, h0, w0, self.n_bins), F32, dev), self.lo, self.hi,
                                          bounded=self.seed_bounded, normalize=self.attr_bounded)
        t = ops.empty((B, h0, w0, self.sp0.cout), dt, dev)
        ops.conv(x0, self.sp0, t, act="relu")
        prev_emb = ops.empty((B, h0, w0, self.emb), dt, dev)
        ops.conv(t, self.sp2, prev_emb)
        emb = prev_emb
        for i, xb in enumerate(x_blocks):
            _, h, w, _ = xb.shape
            p0, p2 = self.proj[i]
            t = ops.empty((B, h, w, p0.cout), dt, dev)
            ops.conv(xb, p0, t, act="relu")
            emb = ops.empty((B, h, w, self.emb), dt, dev)
            ops.conv(t, p2, emb)
            xs = ops.empty((B, h, w, self.emb), dt, dev)
            ops.resize(prev_emb, xs, add=emb)                       # x = emb + up(prev_emb)
            a0, a2 = self.attr[i]
            t = ops.empty((B, h, w, a0.cout), dt, dev)
            ops.conv(xs, a0, t, act="relu")
            A = ops.empty((B, h, w, a2.cout), F32, dev)
            ops.conv(t, a2, A, act="relu" if self.attr_bounded else "softplus")
            b_new = ops.empty((B, h, w, self.n_bins), F32, dev)
            if self.attr_bounded:                                   # attractor.py:100-106: even channels, ReLU + 1e-3
                ops.attractor(A, self.n_attr[i], b_prev, b_new, a_stride=2, a_eps=1e-3, attractor_type=self.attr_type, kind=self.attr_kind)
            else:
                ops.attractor(A, self.n_attr[i], b_prev, b_new, attractor_type=self.attr_type, kind=self.attr_kind)
            b_prev, prev_emb = b_new, emb
            if taps is not None:
                taps[f"bins_centers{i}"] = b_new
        _, H, W, _ = clb.shape
        if self.tail is not None and emb.is_contiguous():
            if self.attr_bounded:
                b_prev = ops.bounded_bin_centers(b_prev, ops.empty(tuple(b_prev.shape), F32, dev), self.lo, self.hi)
            depth = ops.empty((B, H, W), F32, dev)
            ops.bins_tail(clb, emb, self.tail, b_prev, depth, self.min_temp, self.max_temp)
            return depth
        ops.resize(emb, clb[..., 32:32 + self.emb])                 # b_embedding upsampled to (H, W)
        t = ops.empty((B, H, W, self.mlp0.cout), dt, dev)
        ops.conv(clb, self.mlp0, t, act="gelu")
        pt = ops.empty((B, H, W, self.mlp2.cout), F32, dev)
        ops.conv(t, self.mlp2, pt, act="softplus")
        depth = ops.empty((B, H, W), F32, dev)
        if self.attr_bounded:                                       # attractor.py:132-135: the last layer's scaled, sorted, clipped centres
            b_prev = ops.bounded_bin_centers(b_prev, ops.empty(tuple(b_prev.shape), F32, dev), self.lo, self.hi)
        ops.logbinom_depth(pt, b_prev, depth, self.min_temp, self.max_temp)
        return depth


class BranchNet:
    """One ZoeDepth branch with a Depth-Anything (DINOv2 ViT + DPT) core."""

    def __init__(self, sd, prefix, bcfg, process_shape, dtype, device):
        self.dtype, self.device = dtype, device
        enc = bcfg["midas_model_type"]
        self.D, self.depth, self.heads = VIT_ARCH[enc]
        self.C, self.oc = DPT_ARCH[enc]
        self.H, self.W = process_shape
        self.th, self.tw = self.H // 14, self.W // 14
        v = prefix + "core.core.pretrained."
        h = prefix + "core.core.depth_head."
        dev = device

        def pc(name, bias=True, **kw):
            return pk.pack_conv(sd[name + ".weight"], sd[name + ".bias"] if bias else None, dtype=dtype, **kw).to(dev)

        w = sd[v + "patch_embed.proj.weight"]
        self.pe = pk.pack_conv(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), sd[v + "patch_embed.proj.bias"], dtype=dtype,
                               cin_total=592).to(dev)
        self.pos = pk.vit_pos_embed(sd[v + "pos_embed"], self.th, self.tw).to(dev)
        self.cls = _g(sd, v + "cls_token", dev).reshape(-1)
        # float32 compute: the encoder's four linear layers per block run as split-precision GEMMs on the bf16 matrix cores (x = h + m + l,
        # six partial products, float32 accumulation: csrc/gemm_split3.hip; float32-grade error, measured against float64 in
        # tests/op_checks.py gemm_split3).  PF_LINEAR_SPLIT3=0 keeps them on the float32 MFMA.
        self.split3 = dtype == F32 and linear_split3_enabled()
        if self.split3:
            def pc(name, bias=True, _pc=pc, **kw):           # (only the block linears below are re-routed)
                if ".blocks." in name:
                    return pk.pack_conv_split3(sd[name + ".weight"], sd[name + ".bias"] if bias else None, **kw).to(dev)
                return _pc(name, bias, **kw)
        self.blocks = []
        for i in range(self.depth):
            b = f"{v}blocks.{i}."
            self.blocks.append(dict(
                n1=(_g(sd, b + "norm1.weight", dev), _g(sd, b + "norm1.bias", dev)),
                qkv=pc(b + "attn.qkv"),
                proj=pc(b + "attn.proj", scale=sd[b + "ls1.gamma"]),
                n2=(_g(sd, b + "norm2.weight", dev), _g(sd, b + "norm2.bias", dev)),
                fc1=pc(b + "mlp.fc1"),
                fc2=pc(b + "mlp.fc2", scale=sd[b + "ls2.gamma"])))
        self.norm = (_g(sd, v + "norm.weight", dev), _g(sd, v + "norm.bias", dev))
        self.projects = [pc(f"{h}projects.{i}") for i in range(4)]
        self.up4 = pk.pack_conv_transpose(sd[h + "resize_layers.0.weight"], sd[h + "resize_layers.0.bias"], dtype=dtype).to(dev)
        self.up2 = pk.pack_conv_transpose(sd[h + "resize_layers.1.weight"], sd[h + "resize_layers.1.bias"], dtype=dtype).to(dev)
        self.down2 = pc(h + "resize_layers.3")
        self.rn = [pc(f"{h}scratch.layer{i + 1}_rn", bias=False) for i in range(4)]
        self.refine = {}
        for i in range(1, 5):
            r = f"{h}scratch.refinenet{i}."
            self.refine[i] = dict(out=pc(r + "out_conv"),
                                  u1=(pc(r + "resConfUnit1.conv1"), pc(r + "resConfUnit1.conv2")),
                                  u2=(pc(r + "resConfUnit2.conv1"), pc(r + "resConfUnit2.conv2")))
        self.oc1 = pc(h + "scratch.output_conv1")
        self.oc2a = pc(h + "scratch.output_conv2.0")
        self.oc2b = pk.pack_conv(sd[h + "scratch.output_conv2.2.weight"], sd[h + "scratch.output_conv2.2.bias"], dtype=dtype).to(dev)
        # widen the 1-channel relative-depth conv to 8 stored channels so it fills the CLB buffer's
        # padded tail [160:168) (zero weights / zero bias / ReLU -> zeros) without a memset
        self.oc2b.cout = 8
        self.conv2 = pc(prefix + "conv2")
        self.head = BinsHead(sd, prefix, self.C, bcfg, dtype, dev, with_rel=True)

    # ---- pieces ----
    def _rcu(self, ops, x, unit, extra_res=None):
        """x + conv2(relu(conv1(relu(x)))) [+ extra_res]  (blocks.py:69-92; ReLUs are non in-place)"""
        c1, c2 = unit
        t = ops.empty(x.shape[:3] + (c1.cout,), self.dtype, self.device)
        ops.conv(x, c1, t, pad=1, act="relu", relu_in=True)
        y = ops.empty(x.shape[:3] + (c2.cout,), self.dtype, self.device)
        ops.conv(t, c2, y, pad=1, res=x, res2=extra_res)
        return y

    def _refine(self, ops, idx, x, skip, size):
        r = self.refine[idx]
        if skip is not None:
            x = self._rcu(ops, skip, r["u1"], extra_res=x)          # xs[0] + RCU1(xs[1])
        x = self._rcu(ops, x, r["u2"])
        B, _, _, Cc = x.shape
        u = ops.empty((B, size[0], size[1], Cc), self.dtype, self.device)
        ops.resize(x, u)
        y = ops.empty((B, size[0], size[1], r["out"].cout), self.dtype, self.device)
        ops.conv(u, r["out"], y)
        return y

    def vit(self, ops, img, taps=None):
        dt, dev = self.dtype, self.device
        B = img.shape[0]
        T, S, D = self.th * self.tw, self.th * self.tw + 1, self.D
        col = ops.empty((B * T, 592), dt, dev)
        ops.patch_im2col(img, col)
        emb = ops.empty((B * T, D), dt, dev)
        ops.conv(col, self.pe, emb)
        tok = ops.empty((B, S, D), dt, dev)
        ops.assemble_tokens(emb, tok, self.cls, self.pos)
        x = tok.view(B * S, D)
        if taps is not None:
            taps["vit_tokens_in"] = tok.clone()
        feats = []
        if self.split3:                                            # operands of the split GEMMs travel as three bf16 planes ...
            if pk.split3_kmajor_enabled():                         # ... chunk-major [3, K/32, rows, 32]: whole-cache-line DMA pieces (csrc/gemm_split3.hip)
                hbuf, att, mid = (ops.empty((3, n // 32, B * S, 32), torch.bfloat16, dev) for n in (D, D, 4 * D))
            else:
                hbuf, att, mid = (ops.empty((3, B * S, n), torch.bfloat16, dev) for n in (D, D, 4 * D))
            # ... and so do q / k / v (row-major: the attention kernel reads token rows): attention in split precision (csrc/vit.hip)
            qkv = ops.empty((3, B * S, 3 * D), torch.bfloat16, dev) if attention_split3_enabled() else ops.empty((B * S, 3 * D), dt, dev)
        else:
            hbuf = ops.empty((B * S, D), dt, dev)
            qkv = ops.empty((B * S, 3 * D), dt, dev)
            att = ops.empty((B * S, D), dt, dev)
            mid = ops.empty((B * S, 4 * D), dt, dev)
        for i, blk in enumerate(self.blocks):
            if self.split3:
                ops.layernorm_split3(x, hbuf, blk["n1"][0], blk["n1"][1], 1e-6)
                ops.conv_split3(hbuf, blk["qkv"], qkv)
                ops.vit_attention(qkv, att, B, S, self.heads)
                ops.conv_split3(att, blk["proj"], x, res=x)
                ops.layernorm_split3(x, hbuf, blk["n2"][0], blk["n2"][1], 1e-6)
                ops.conv_split3(hbuf, blk["fc1"], mid, act="gelu")
                ops.conv_split3(mid, blk["fc2"], x, res=x)
            else:
                ops.layernorm(x, hbuf, blk["n1"][0], blk["n1"][1], 1e-6)
                ops.conv(hbuf, blk["qkv"], qkv)
                ops.vit_attention(qkv, att, B, S, self.heads)
                ops.conv(att, blk["proj"], x, res=x)                   # x += ls1 * proj(attn)
                ops.layernorm(x, hbuf, blk["n2"][0], blk["n2"][1], 1e-6)
                ops.conv(hbuf, blk["fc1"], mid, act="gelu")
                ops.conv(mid, blk["fc2"], x, res=x)                    # x += ls2 * fc2(gelu(fc1))
            if taps is not None and i in (0, self.depth - 1):
                taps[f"vit_block{i}"] = x.view(B, S, D).clone()
            if i >= self.depth - 4:
                f = ops.empty((B, self.th, self.tw, D), dt, dev)
                ops.layernorm(x, f.view(B * T, D), self.norm[0], self.norm[1], 1e-6, batches=B, in_rows_per_batch=S,
                              in_row_offset=1, out_rows_per_batch=T)
                feats.append(f)
        return feats

    def forward(self, ops, img, taps=None, vit_feats=None):
        """img [B,3,H,W] float32 in [0,1] -> (depth f32 [B,H,W], feats low->high
        [x_d0, r4, r3, r2, r1, out_conv(32)]).  ``vit_feats``: the four encoder outputs of these B images when the caller has already
        run :meth:`vit` on a larger batch (all tiles of an image in one launch per layer: fewer, better-filled GEMM launches)."""
        dt, dev = self.dtype, self.device
        B = img.shape[0]
        th, tw, C = self.th, self.tw, self.C
        feats = self.vit(ops, img, taps) if vit_feats is None else vit_feats
        if taps is not None:
            for i, f in enumerate(feats):
                taps[f"vit_out{i}"] = f
        maps = []
        for i, f in enumerate(feats):
            p = ops.empty((B, th, tw, self.projects[i].cout), dt, dev)
            ops.conv(f, self.projects[i], p)
            if i == 0:
                y = ops.empty((B, th * 4, tw * 4, self.oc[0]), dt, dev)
                ops.conv(p, self.up4, y)
            elif i == 1:
                y = ops.empty((B, th * 2, tw * 2, self.oc[1]), dt, dev)
                ops.conv(p, self.up2, y)
            elif i == 2:
                y = p
            else:
                y = ops.empty((B, (th + 1) // 2, (tw + 1) // 2, self.oc[3]), dt, dev)
                ops.conv(p, self.down2, y, stride=2, pad=1)
            maps.append(y)
        rn = []
        for i in range(4):
            y = ops.empty(maps[i].shape[:3] + (C,), dt, dev)
            ops.conv(maps[i], self.rn[i], y, pad=1)
            rn.append(y)
        r4 = self._refine(ops, 4, rn[3], None, rn[2].shape[1:3])
        r3 = self._refine(ops, 3, r4, rn[2], rn[1].shape[1:3])
        r2 = self._refine(ops, 2, r3, rn[1], rn[0].shape[1:3])
        r1 = self._refine(ops, 1, r2, rn[0], (rn[0].shape[1] * 2, rn[0].shape[2] * 2))
        o1 = ops.empty(r1.shape[:3] + (self.oc1.cout,), dt, dev)
        ops.conv(r1, self.oc1, o1, pad=1)
        H, W = th * 14, tw * 14
        o1u = ops.empty((B, H, W, self.oc1.cout), dt, dev)
        ops.resize(o1, o1u)
        clb = self.head.new_clb_buffer(ops, B, H, W)
        out_conv = clb[..., 0:32]
        ops.conv(o1u, self.oc2a, out_conv, pad=1, act="relu")        # hook 'out_conv'
        ops.conv(out_conv, self.oc2b, clb[..., 32 + self.head.emb:], act="relu")   # relative depth (+ zero tail)
        x_d0 = ops.empty(rn[3].shape[:3] + (C,), dt, dev)
        ops.conv(rn[3], self.conv2, x_d0)
        if taps is not None:
            taps.update(dpt_layer1_rn=rn[0], dpt_layer4_rn=rn[3], rel_depth=clb[..., 32 + self.head.emb:33 + self.head.emb])
        depth = self.head.run(ops, x_d0, [r4, r3, r2, r1], clb, taps)
        return depth, [x_d0, r4, r3, r2, r1, out_conv]


class ExternalCoreBranchNet:
    """ZoeDepth branch whose relative-depth core is EXTERNAL (type 'ZoeDepth': MiDaS DPT_BEiT_L_384, BASELINE configs[4]).

    The MiDaS/BEiT core is not in the reference tree (torch.hub repo AyaanShah2204/MiDaS, no commit pinned, midas.py:340), so
    it is supplied as a feature provider -- the reference's own injection point ``ZoeDepth.forward(hack_feature=(rel_depth,
    out))`` (zoedepth_v1.py:160-166):
        provider(img [B,3,H,W] float32 in [0,1]) -> (rel_depth [B,H,W] f32, [btlnck, x_block0..3, out_conv] NCHW f32)
    with btlnck at H/32 and the blocks at H/16 ... H/2 (256 channels), out_conv 32 channels at HxW.  Everything after the core --
    conv2 and the metric-bins head (zoedepth_v1.py:170-219) -- runs on the HIP op set exactly like the Depth-Anything branch."""

    def __init__(self, sd, prefix, bcfg, process_shape, dtype, device, provider):
        if provider is None:
            raise NotImplementedError(
                "branch type 'ZoeDepth' needs a relative-depth core: the MiDaS/BEiT encoder is an un-vendored torch.hub "
                "repository (midas.py:340).  Pass core_providers=(coarse, fine) to PatchFusion or call set_core_providers().")
        self.dtype, self.device, self.provider = dtype, device, provider
        self.C = branch_channels(bcfg)
        self.H, self.W = process_shape
        self.conv2 = pk.pack_conv(sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], dtype=dtype).to(device)
        self.head = BinsHead(sd, prefix, self.C, bcfg, dtype, device, with_rel=True)

    def forward(self, ops, img, taps=None):
        dt, dev = self.dtype, self.device
        rel, feats = self.provider(img)
        assert len(feats) == 6, "provider must return [btlnck, x_block0..3, out_conv]"
        nhwc = [f.detach().to(dev).permute(0, 2, 3, 1).contiguous().to(dt) for f in feats]      # re-layout only
        btl, blocks, outc = nhwc[0], nhwc[1:5], nhwc[5]
        B, H, W = rel.shape
        clb = self.head.new_clb_buffer(ops, B, H, W)            # [last 0..31 | emb 32..159 | rel 160 (+7 zero pad)]
        ops.copy_channels(outc, clb[..., :32])
        tail = torch.zeros((B, H, W, 8), dtype=dt, device=dev)
        tail[..., 0] = rel.detach().to(device=dev, dtype=dt)
        clb[..., 32 + self.head.emb:] = tail
        x_d0 = ops.empty(btl.shape[:3] + (self.C,), dt, dev)
        ops.conv(btl, self.conv2, x_d0)
        depth = self.head.run(ops, x_d0, blocks, clb, taps)
        return depth, [x_d0] + blocks + [clb[..., :32]]


class G2LNet:
    """The six global-to-local Swin stacks; input = whole-image coarse pyramid, i.e. patch invariant."""
    DEPTH = [4, 4, 3, 3, 2, 2]     # guided_fusion_model.py:109-110 defaults, reversed at :141-143
    HEADS = [32, 32, 16, 16, 8, 8]

    def __init__(self, sd, gcfg, dtype, device, prefix="guided_fusion."):
        self.dtype, self.device = dtype, device
        from .spec import GF_DEFAULT_IN_CHANNELS
        ch = list(gcfg.get("in_channels", GF_DEFAULT_IN_CHANNELS))[::-1]
        self.levels = []
        for i, C in enumerate(ch):
            g = f"{prefix}g2l_list.{i}."
            blocks = []
            for j in range(self.DEPTH[i]):
                b = f"{g}g2l_layer.blocks.{j}."

                def pc(name):
                    return pk.pack_conv(sd[b + name + ".weight"], sd[b + name + ".bias"], dtype=dtype).to(device)
                blocks.append(dict(n1=(_g(sd, b + "norm1.weight", device), _g(sd, b + "norm1.bias", device)),
                                   n2=(_g(sd, b + "norm2.weight", device), _g(sd, b + "norm2.bias", device)),
                                   qkv=pc("attn.qkv"), proj=pc("attn.proj"), fc1=pc("mlp.fc1"), fc2=pc("mlp.fc2"),
                                   bias_table=_g(sd, b + "attn.relative_position_bias_table", device),
                                   shift=0 if j % 2 == 0 else 6))
            self.levels.append(dict(C=C, heads=self.HEADS[i], blocks=blocks,
                                    ape=_g(sd, g + "absolute_pos_embed", device)[0],
                                    norm=(_g(sd, g + "g2l_layer_norm.weight", device), _g(sd, g + "g2l_layer_norm.bias", device))))

    def forward_level(self, ops, i, feat):
        dt, dev = self.dtype, self.device
        lv = self.levels[i]
        B, H, W, C = feat.shape
        Hp, Wp = int(math.ceil(H / 12)) * 12, int(math.ceil(W / 12)) * 12
        nt = B * Hp * Wp
        x = ops.empty((B, H, W, C), dt, dev)
        ops.copy_channels(feat, x)
        ops.add_rowwise(x.view(B, H * W, C), lv["ape"])
        xw = ops.empty((nt, C), dt, dev)
        qkv = ops.empty((nt, 3 * C), dt, dev)
        ao = ops.empty((nt, C), dt, dev)
        pr = ops.empty((nt, C), dt, dev)
        hbuf = ops.empty((B * H * W, C), dt, dev)
        mid = ops.empty((B * H * W, 4 * C), dt, dev)
        for blk in lv["blocks"]:
            ops.swin_ln_partition(x, xw, blk["n1"][0], blk["n1"][1], 1e-5, blk["shift"])
            ops.conv(xw, blk["qkv"], qkv)
            ops.swin_window_attention(qkv, ao, blk["bias_table"], B, Hp, Wp, C, lv["heads"], blk["shift"])
            ops.conv(ao, blk["proj"], pr)
            x2 = ops.empty((B, H, W, C), dt, dev)
            ops.swin_unpartition_add(pr, x, x2, blk["shift"])
            x = x2
            xf = x.view(B * H * W, C)
            ops.layernorm(xf, hbuf, blk["n2"][0], blk["n2"][1], 1e-5)
            ops.conv(hbuf, blk["fc1"], mid, act="gelu")
            ops.conv(mid, blk["fc2"], xf, res=xf)
        out = ops.empty((B, H, W, C), dt, dev)
        ops.layernorm(x.view(B * H * W, C), out.view(B * H * W, C), lv["norm"][0], lv["norm"][1], 1e-5)
        return out

    def forward(self, ops, coarse_feats):
        return [self.forward_level(ops, i, f) for i, f in enumerate(coarse_feats)]


class FusionNet:
    """fusion_conv_list + GuidedFusionPatchFusion U-Net + the fusion-side metric-bins head."""

    def __init__(self, sd, cfg, dtype, device):
        self.dtype, self.device = dtype, device
        fb = cfg["fine_branch"]
        self.C = branch_channels(fb)
        self.ps = tuple(cfg["patch_process_shape"])
        self.sizes = pyramid_sizes(self.ps, fb["type"])[::-1]       # L0..L5 (h, w)
        self.ch = [self.C] * 5 + [32]                               # channels per level L0..L5
        g = "guided_fusion."

        def pc(name, **kw):
            return pk.pack_conv(sd[name + ".weight"], sd.get(name + ".bias"), dtype=dtype, **kw).to(device)

        def bn(name):
            return tuple(sd[f"{name}.{k}"] for k in ("weight", "bias", "running_mean", "running_var"))

        self.fconv = [pc(f"fusion_conv_list.{i}") for i in range(6)]

        def dcbn(name, **kw):
            return (pc(name + ".double_conv.0", bn=bn(name + ".double_conv.1"), **kw),
                    pc(name + ".double_conv.3", bn=bn(name + ".double_conv.4")))

        def dcwobn(name):
            return (pc(name + ".double_conv.0"), pc(name + ".double_conv.2"))

        self.inc = dcbn(g + "inc", cin_total=8)
        self.down = [dcbn(f"{g}down_conv_list.{i}.maxpool_conv.1") for i in range(5)]
        self.upc = [dcwobn(f"{g}up_conv_list.{i}.conv") for i in range(5)]
        self.convs = [dcwobn(f"{g}convs.{i}") for i in range(6)]
        self.head = BinsHead(sd, "", self.C, cfg["coarse_branch"], dtype, device, with_rel=False,
                             min_depth=cfg["min_depth"], max_depth=cfg["max_depth"])

    def forward(self, ops, crops, rois, fine_depth, fine_feats, coarse_depth, coarse_feats, g2l, taps=None):
        """crops [B,3,H,W] f32; rois f32 [B,5] (batch index 0, box in process coordinates);
        fine_depth f32 [B,H,W]; coarse_depth f32 [1,1,H,W]; returns depth f32 [B,H,W]."""
        dt, dev = self.dtype, self.device
        B = crops.shape[0]
        H, W = self.ps
        ch, sizes = self.ch, self.sizes
        # --- coarse ROIs + fusion convs (patchfusion.py:240-267) ---
        cd_roi = ops.empty((B, 1, H, W), F32, dev)
        ops.roi_align_depth(coarse_depth, rois, cd_roi, 1.0)
        guide = []
        for i in range(6):
            h, w = sizes[i]
            cat = ops.empty((B, h, w, 2 * ch[i]), dt, dev)
            ops.roi_align(coarse_feats[i], rois, cat[..., :ch[i]], h / H)
            ops.copy_channels(fine_feats[i], cat[..., ch[i]:])
            gc = ops.empty((B, h, w, ch[i]), dt, dev)
            ops.conv(cat, self.fconv[i], gc, pad=1)
            guide.append(gc)
        # --- encoder (guided_fusion_model.py:178-184) ---
        inp = ops.empty((B, H, W, 8), dt, dev)
        ops.pack_fusion_input(cd_roi, fine_depth, crops, inp)
        u5 = ops.empty((B, H, W, ch[5] + 2 * ch[4]), dt, dev)       # Upv1 concat buffer of the last level
        t = ops.empty((B, H, W, 32), dt, dev)
        ops.conv(inp, self.inc[0], t, pad=1, act="relu")
        e = u5[..., :32]
        ops.conv(t, self.inc[1], e, pad=1, act="relu")
        enc = [e]
        for i in range(5):
            _, hh, ww, cc = enc[-1].shape
            mp = ops.empty((B, hh // 2, ww // 2, cc), dt, dev)
            ops.maxpool2(enc[-1], mp)
            c0, c1 = self.down[i]
            t = ops.empty((B, hh // 2, ww // 2, c0.cout), dt, dev)
            ops.conv(mp, c0, t, pad=1, act="relu")
            e = ops.empty((B, hh // 2, ww // 2, c1.cout), dt, dev)
            ops.conv(t, c1, e, pad=1, act="relu")
            enc.append(e)
        enc = enc[::-1]
        # --- decoder (guided_fusion_model.py:186-205) ---
        clb = self.head.new_clb_buffer(ops, B, H, W)
        fused, temp = [], None
        for i in range(6):
            h, w = sizes[i]
            c = ch[i]
            v = ops.empty((B, h, w, 2 * c), dt, dev)                 # cat[feat_enc, roi_align(g2l)]
            if i == 0:
                ops.resize(enc[0], v[..., :c])
            else:
                cp = ch[i - 1]
                u = u5 if i == 5 else ops.empty((B, h, w, c + 2 * cp), dt, dev)
                # Upv1 concat buffer [enc resized to the DPT grid | up(temp) | up(guide)] in ONE launch (whole rows of u);
                # at the last level enc already sits in u5[..., :32] (the inc conv wrote it there)
                if i < 5:
                    ops.resize_concat([enc[i], temp, guide[i - 1]], u)
                else:
                    ops.resize_concat([temp, guide[i - 1]], u[..., c:])
                c0, c1 = self.upc[i - 1]
                t = ops.empty((B, h, w, c0.cout), dt, dev)
                ops.conv(u, c0, t, pad=1, act="relu")
                ops.conv(t, c1, v[..., :c], pad=1, act="relu")
            ops.roi_align(g2l[i], rois, v[..., c:], h / H)
            c0, c1 = self.convs[i]
            t = ops.empty((B, h, w, c0.cout), dt, dev)
            ops.conv(v, c0, t, pad=1, act="relu")
            temp = clb[..., :32] if i == 5 else ops.empty((B, h, w, c1.cout), dt, dev)
            ops.conv(t, c1, temp, pad=1, act="relu")
            fused.append(temp)
            if taps is not None:
                taps[f"gf_out{i}"] = temp
        return self.head.run(ops, fused[0], fused[1:5], clb, taps)
