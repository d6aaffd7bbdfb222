"""TEST INFRASTRUCTURE ONLY.  A plain-PyTorch (fp32 math) implementation of the op set that
patchfusion_amd.hip_ops exposes, with identical signatures and buffer conventions.

Two uses:
  * ``-m "not gpu"``: drive patchfusion_amd.engine on CPU to validate wiring + weight packing against the
    oracle (no HIP kernel involved -- this is NOT a product path; the product only accepts hip_ops);
  * ``-m gpu``: the per-op reference each hand-written HIP kernel is compared against.
"""
import os

import torch
import torch.nn.functional as F

from oracle import third_party as tp
from patchfusion_amd.packing import kmajor_to_rows, rows_to_kmajor, split3 as pk_split3, unpack_conv, winograd_applies


def _as4(t):
    if t.dim() == 2:
        return t.unsqueeze(0).unsqueeze(0)
    if t.dim() == 3:
        return t.unsqueeze(0)
    return t


def _act(v, act):
    if act in (None, "none"):
        return v
    if act == "relu":
        return F.relu(v)
    if act == "gelu":
        return F.gelu(v)
    if act == "softplus":
        return F.softplus(v)
    raise ValueError(act)


def _conv_ref(xin, w, stride, pad):
    """fp32 convolution reference.  On the GPU the 1x1 and the 3x3 / stride-1 cases are evaluated as plain fp32 matmuls
    (one per filter tap over the zero-padded input): same arithmetic as F.conv2d, but MIOpen's per-shape solver search /
    kernel compilation (minutes on a fresh box for the big f32 shapes) stays out of the test run."""
    KH, KW = w.shape[-2:]
    if xin.is_cuda and stride == 1 and ((KH, KW, pad) == (1, 1, 0) or (KH, KW, pad) == (3, 3, 1)):
        B, C, H, W = xin.shape
        xp = F.pad(xin.permute(0, 2, 3, 1), (0, 0, pad, pad, pad, pad))            # NHWC, zero border
        out = None
        for ky in range(KH):
            for kx in range(KW):
                t = torch.matmul(xp[:, ky:ky + H, kx:kx + W, :], w[:, :, ky, kx].t())
                out = t if out is None else out + t
        return out.permute(0, 3, 1, 2)
    return F.conv2d(xin, w, None, stride=stride, padding=pad)


def _conv_winograd_ref(xin, pw):
    """csrc/winograd.hip in float32 torch: V = B^T d B per tile, M = U . V per transform point, Y = A^T M A (cropped to H x W)"""
    from patchfusion_amd.packing import WINO_AT, WINO_BT
    m = pw.wino_m
    a = m + 2
    Bt = torch.tensor(WINO_BT[m], dtype=torch.float32, device=xin.device)
    At = torch.tensor(WINO_AT[m], dtype=torch.float32, device=xin.device)
    B, C, H, W = xin.shape
    TH, TW = -(-H // m), -(-W // m)
    xp = F.pad(xin, (1, TW * m - W + 1, 1, TH * m - H + 1))
    t = xp.unfold(2, a, m).unfold(3, a, m)                                            # [B, C, TH, TW, a, a]
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, t, Bt)
    U = pw.wino_u.to(xin.device)[:, :pw.cout, :C].reshape(a, a, pw.cout, C)
    M = torch.einsum("iloc,bcyxil->boyxil", U, V)
    Y = torch.einsum("pi,boyxil,ql->boyxpq", At, M, At)                               # [B, N, TH, TW, m, m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, pw.cout, TH * m, TW * m)[:, :, :H, :W]


class FakeOps:
    name = "fake"

    @staticmethod
    def empty(shape, dtype, device):
        # NaN-fill so that reads of never-written channels are caught by the wiring tests
        return torch.full(shape, float("nan"), dtype=dtype, device=device)

    @staticmethod
    def zeros(shape, dtype, device):
        return torch.zeros(shape, dtype=dtype, device=device)

    @staticmethod
    def conv(x, pw, y, stride=1, pad=0, act=None, relu_in=False, res=None, res2=None, _direct=None):
        x4, y4 = _as4(x), _as4(y)
        w = unpack_conv(pw).to(x4.device)
        xin = x4[..., :pw.cin].float().permute(0, 3, 1, 2)
        if relu_in:
            xin = F.relu(xin)
        s = pw.shuffle
        if (os.environ.get("PF_FAKE_WINOGRAD") == "1" and not _direct and not xin.is_cuda and pw.wino_u is not None and
                winograd_applies(pw, xin.shape[0] * xin.shape[2] * xin.shape[3], stride, pad, act)):
            # opt-in for the CPU wiring tests of the Winograd path (tests/test_winograd_cpu.py, one end-to-end case): the three steps
            # with the PACKED filters.  Everywhere else -- and always on the GPU, where this class is the CHECKER of the HIP kernels --
            # the direct convolution is evaluated.
            v = _conv_winograd_ref(xin, pw)
        else:
            v = _conv_ref(xin, w, stride, pad)                                       # [B, N, OH, OW]
        if s > 1:
            B, N, OH, OW = v.shape
            ct = N // (s * s)
            v = v.view(B, s, s, ct, OH, OW).permute(0, 3, 4, 1, 5, 2).reshape(B, ct, OH * s, OW * s)
        n = v.shape[1]
        if pw.bias is not None:
            v = v + pw.bias[:n].to(v.device).view(1, -1, 1, 1)
        v = _act(v, act)
        if pw.scale is not None:
            v = v * pw.scale[:n].to(v.device).view(1, -1, 1, 1)
        v = v.permute(0, 2, 3, 1)
        if res is not None:
            v = v + _as4(res)[..., :n].float()
        if res2 is not None:
            v = v + _as4(res2)[..., :n].float()
        y4[..., :n] = v.to(y4.dtype)
        return y

    @staticmethod
    def bins_tail(clb, emb, tw, centers, depth, min_temp, max_temp):
        """the four steps the fused kernel replaces, with the packed layers it was built from"""
        B, H, W, _ = clb.shape
        buf = clb.clone()
        FakeOps.resize(emb, buf[..., 32:32 + emb.shape[-1]])
        t = torch.zeros((B, H, W, tw.mlp0.cout), dtype=torch.float32, device=clb.device)
        FakeOps.conv(buf, tw.mlp0, t, act="gelu")
        pt = torch.zeros((B, H, W, tw.mlp2.cout), dtype=torch.float32, device=clb.device)
        FakeOps.conv(t, tw.mlp2, pt, act="softplus")
        FakeOps.logbinom_depth(pt, centers, depth, min_temp, max_temp)

    # ---- split-precision linears: the planes are summed back to float32 (exact) and the layer evaluated in plain float32 ----
    @staticmethod
    def _store3(y3, v):
        h, m, l = pk_split3(v)
        if y3.dim() == 4:                                    # chunk-major planes [3, cols/32, rows, 32] (packing.rows_to_kmajor)
            y3[:] = rows_to_kmajor(torch.stack([h, m, l]))
        else:
            y3[0], y3[1], y3[2] = h, m, l

    @staticmethod
    def _rows(t3):
        """three planes, row-major [3, rows, cols] or chunk-major [3, cols/32, rows, 32] -> float32 [rows, cols] (exact sum)"""
        return (kmajor_to_rows(t3) if t3.dim() == 4 else t3).float().sum(0)

    @staticmethod
    def split3(x, y3):
        FakeOps._store3(y3, x.float())
        return y3

    @staticmethod
    def conv_split3(x3, pw, y, act=None, res=None, res2=None):
        x = FakeOps._rows(x3)
        w = FakeOps._rows(pw.w)[:pw.cout, :pw.cin].to(x.device)
        v = x[:, :pw.cin] @ w.t()
        if pw.bias is not None:
            v = v + pw.bias[:pw.cout].to(v.device)
        v = _act(v, act)
        if pw.scale is not None:
            v = v * pw.scale[:pw.cout].to(v.device)
        if res is not None:
            v = v + res[:, :pw.cout].float()
        if res2 is not None:
            v = v + res2[:, :pw.cout].float()
        if y.dtype == torch.bfloat16:
            FakeOps._store3(y if y.dim() == 4 else y[:, :, :pw.cout], v)
        else:
            y[:, :pw.cout] = v
        return y

    @staticmethod
    def layernorm_split3(x, y3, g, b, eps):
        FakeOps._store3(y3, F.layer_norm(x.float(), (x.shape[-1],), g, b, eps))
        return y3

    @staticmethod
    def patch_im2col(img, out):
        B, _, H, W = img.shape
        mean = torch.tensor([0.485, 0.456, 0.406], device=img.device).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], device=img.device).view(1, 3, 1, 1)
        x = (img - mean) / std
        th, tw = H // 14, W // 14
        p = x.view(B, 3, th, 14, tw, 14).permute(0, 2, 4, 3, 5, 1).reshape(B * th * tw, 588)
        out[:, :588] = p.to(out.dtype)
        out[:, 588:] = 0

    @staticmethod
    def assemble_tokens(emb, tokens, cls, pos):
        B, S, D = tokens.shape
        tokens[:, 0] = (cls + pos[0]).to(tokens.dtype)
        tokens[:, 1:] = (emb.view(B, S - 1, D).float() + pos[1:]).to(tokens.dtype)

    @staticmethod
    def layernorm(x, y, g, b, eps, batches=1, in_rows_per_batch=None, in_row_offset=0, out_rows_per_batch=None):
        D = x.shape[-1]
        xf = x.reshape(-1, D).float()
        if in_rows_per_batch is not None:
            xf = xf.view(batches, in_rows_per_batch, D)[:, in_row_offset:in_row_offset + out_rows_per_batch].reshape(-1, D)
        v = F.layer_norm(xf, (D,), g, b, eps)
        y.view(-1, D)[:] = v.to(y.dtype)

    @staticmethod
    def vit_attention(qkv, out, B, S, heads):
        if qkv.dim() == 3:                                   # three bf16 planes of the QKV output: summed back to float32 (exact)
            qkv = qkv.float().sum(0)
        D = qkv.shape[1] // 3
        q, k, v = qkv.float().view(B, S, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
        a = ((q * (D // heads) ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B * S, D)
        if qkv.dtype == torch.float32 and out.dtype == torch.bfloat16:       # split planes for the projection (ops.conv_split3)
            FakeOps._store3(out, o)
        else:
            out[:] = o.to(out.dtype)

    @staticmethod
    def swin_ln_partition(x, xw, g, b, eps, shift):
        B, H, W, C = x.shape
        v = F.layer_norm(x.float(), (C,), g, b, eps)
        pr, pb = (12 - W % 12) % 12, (12 - H % 12) % 12
        v = F.pad(v, (0, 0, 0, pr, 0, pb))
        if shift > 0:
            v = torch.roll(v, shifts=(-shift, -shift), dims=(1, 2))
        Hp, Wp = H + pb, W + pr
        v = v.view(B, Hp // 12, 12, Wp // 12, 12, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)
        xw[:] = v.to(xw.dtype)

    @staticmethod
    def swin_window_attention(qkv, out, bias_table, B, Hp, Wp, C, heads, shift):
        from oracle.pf_oracle import swin_shift_mask
        from patchfusion_amd.spec import relative_position_index
        nW = qkv.shape[0] // 144
        hd = C // heads
        q, k, v = qkv.float().view(nW, 144, 3, heads, hd).permute(2, 0, 3, 1, 4)
        attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
        idx = relative_position_index().to(qkv.device).view(-1)
        attn = attn + bias_table[idx].view(144, 144, heads).permute(2, 0, 1).unsqueeze(0)
        if shift > 0:
            mask = swin_shift_mask(Hp, Wp, 12, shift, qkv.device)
            attn = attn.view(B, nW // B, heads, 144, 144) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, heads, 144, 144)
        attn = attn.softmax(dim=-1)
        out[:] = (attn @ v).transpose(1, 2).reshape(nW * 144, C).to(out.dtype)

    @staticmethod
    def swin_unpartition_add(proj, shortcut, y, shift):
        B, H, W, C = shortcut.shape
        Hp, Wp = (H + 11) // 12 * 12, (W + 11) // 12 * 12
        v = proj.float().view(B, Hp // 12, Wp // 12, 12, 12, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if shift > 0:
            v = torch.roll(v, shifts=(shift, shift), dims=(1, 2))
        y[..., :C] = (shortcut.float() + v[:, :H, :W]).to(y.dtype)

    @staticmethod
    def add_rowwise(x, pos):
        x[:] = (x.float() + pos.unsqueeze(0)).to(x.dtype)

    @staticmethod
    def resize(x, y, add=None, dtype=None):
        x4, y4 = _as4(x), _as4(y)
        C = x4.shape[-1]
        v = F.interpolate(x4.float().permute(0, 3, 1, 2), size=y4.shape[1:3], mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        if add is not None:
            v = _as4(add)[..., :C].float() + v
        y4[..., :C] = v.to(y4.dtype)

    @staticmethod
    def resize_concat(xs, y):
        c0 = 0
        for x in xs:
            FakeOps.resize(x, y[..., c0:c0 + x.shape[-1]])
            c0 += x.shape[-1]

    @staticmethod
    def crop_resize(img, boxes, out):
        P, _, oh, ow = out.shape
        for p, (x0, y0, x1, y1) in enumerate(boxes.tolist()):
            out[p] = F.interpolate(img[None, :, y0:y1, x0:x1], size=(oh, ow), mode="bilinear", align_corners=True)[0]

    @staticmethod
    def roi_align_depth(feat, rois, y, spatial_scale):
        y[:] = tp.roi_align(feat, rois, y.shape[2:], spatial_scale, aligned=True)

    @staticmethod
    def roi_align(feat, rois, y, spatial_scale, dtype=None):
        C = feat.shape[-1]
        v = tp.roi_align(feat.float().permute(0, 3, 1, 2), rois, y.shape[1:3], spatial_scale, aligned=True)
        y[..., :C] = v.permute(0, 2, 3, 1).to(y.dtype)

    @staticmethod
    def maxpool2(x, y):
        C = x.shape[-1]
        y[..., :C] = F.max_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).to(y.dtype)

    @staticmethod
    def copy_channels(x, y):
        x4, y4 = _as4(x), _as4(y)
        y4[..., :x4.shape[-1]] = x4.to(y4.dtype)

    @staticmethod
    def pack_fusion_input(cdepth, fdepth, crops, y):
        B = y.shape[0]
        y[..., 0] = cdepth.reshape(B, *y.shape[1:3]).to(y.dtype)
        y[..., 1] = fdepth.reshape(B, *y.shape[1:3]).to(y.dtype)
        y[..., 2:5] = crops.permute(0, 2, 3, 1).to(y.dtype)
        y[..., 5:] = 0

    @staticmethod
    def copy_plane(src, dst):
        """dst[...] = src[:, 0] for float32 depth planes (device-to-device, async on the current stream)"""
        dst.copy_(src[:, 0], non_blocking=True)

    @staticmethod
    def nhwc_to_nchw(x, Cc=None):
        Cc = Cc or x.shape[-1]
        return x[..., :Cc].float().permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def attractor(A, n_attr, b_prev, out, a_stride=1, a_eps=0.0, attractor_type="inv", kind="mean"):
        h, w = out.shape[1:3]
        c = F.interpolate(b_prev.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True)   # [B,nb,h,w]
        a = A[..., :n_attr * a_stride:a_stride].permute(0, 3, 1, 2) + a_eps
        dx = a.unsqueeze(2) - c.unsqueeze(1)
        d = torch.exp(-300.0 * dx.abs() ** 2) * dx if attractor_type == "exp" else dx / (1 + 300.0 * dx.pow(2))
        out[:] = (c + (d.mean(dim=1) if kind == "mean" else d.sum(dim=1))).permute(0, 2, 3, 1)

    @staticmethod
    def seed_bin_centers(x, out, min_depth, max_depth, bounded, normalize):
        nb = out.shape[-1]
        c = x[..., :nb].float()
        if bounded:
            Bn = c + 1e-3
            widths = (max_depth - min_depth) * (Bn / Bn.sum(dim=-1, keepdim=True))
            edges = torch.cumsum(torch.cat([torch.full_like(widths[..., :1], min_depth), widths], dim=-1), dim=-1)
            c = 0.5 * (edges[..., :-1] + edges[..., 1:])
        if normalize:
            c = (c - min_depth) / (max_depth - min_depth)
        out[:] = c
        return out

    @staticmethod
    def bounded_bin_centers(b, out, min_depth, max_depth):
        out[:] = torch.clip(torch.sort((max_depth - min_depth) * b + min_depth, dim=-1)[0], min_depth, max_depth)
        return out

    @staticmethod
    def logbinom_depth(pt, centers, depth, min_temp, max_temp):
        from oracle.pf_oracle import up
        B, h, w = depth.shape
        nb = centers.shape[-1]
        q = pt[..., :4] + 1e-4
        p = q[..., 0] / (q[..., 0] + q[..., 1])
        t = q[..., 2] / (q[..., 2] + q[..., 3])
        t = ((max_temp - min_temp) * t + min_temp).unsqueeze(1)
        x = p.unsqueeze(1)
        om = torch.clamp(1 - x, 1e-4, 1)
        x = torch.clamp(x, 1e-4, 1)
        k = torch.arange(nb, device=pt.device, dtype=torch.float32).view(1, -1, 1, 1)
        eps = 1e-7
        n_, k_ = (nb - 1) + eps, k + eps
        logc = n_ * torch.log(torch.tensor(n_)) - k_ * torch.log(k_) - (n_ - k_) * torch.log(n_ - k_ + eps)
        yk = logc + k * torch.log(x) + (nb - 1 - k) * torch.log(om)
        probs = torch.softmax(yk / t, dim=1)
        c = up(centers.permute(0, 3, 1, 2), (h, w))
        depth[:] = (probs * c).sum(dim=1)

    @staticmethod
    def stitch_init(pred, count, depth, mask, yx):
        P, ph, pw = depth.shape
        for p, (y0, x0) in enumerate(yx.tolist()):
            pred[y0:y0 + ph, x0:x0 + pw] = depth[p] * mask
            count[y0:y0 + ph, x0:x0 + pw] = mask

    @staticmethod
    def stitch_finish_init(avg, pred, count):
        avg[:] = pred / count

    @staticmethod
    def stitch_update(avg, count, depth, mask, y0, x0):
        ph, pw = mask.shape
        d = depth
        if tuple(depth.shape) != (ph, pw):
            d = F.interpolate(depth[None, None], (ph, pw))[0, 0]
        a = avg[y0:y0 + ph, x0:x0 + pw]
        c = count[y0:y0 + ph, x0:x0 + pw]
        avg[y0:y0 + ph, x0:x0 + pw] = (d * mask + c * a) / (c + mask)
        count[y0:y0 + ph, x0:x0 + pw] = c + mask

    @staticmethod
    def resize_nearest_f32(x, y):
        y[:] = F.interpolate(x[None, None], size=tuple(y.shape))[0, 0]

    @staticmethod
    def resize_bilinear_f32(x, y):
        y[:] = F.interpolate(x[None, None], size=tuple(y.shape), mode="bilinear", align_corners=True)[0, 0]

    # ---------------- input / output side (reference semantics through torch / the oracle) ----------------
    @staticmethod
    def u8_bicubic_to_f32(img_u8, out, reverse_channels=False):
        x = (img_u8.double() / 255.0).permute(2, 0, 1)[None]
        if tuple(x.shape[-2:]) != tuple(out.shape[-2:]):
            x = F.interpolate(x, tuple(out.shape[-2:]), mode="bicubic", align_corners=True)
        x = x[0].float()
        out[:] = x.flip(0) if reverse_channels else x
        return out

    @staticmethod
    def percentiles(x, q0, q1, invalid_val=None, out=None, invalid_mask=None):
        from oracle import io_oracle
        v = x.cpu().numpy().ravel()
        if invalid_mask is not None:
            v = v[invalid_mask.cpu().numpy().ravel() == 0]
        elif invalid_val is not None:
            v = v[v != invalid_val]
        r = torch.tensor([float(io_oracle.percentile_linear(v, q0)), float(io_oracle.percentile_linear(v, q1))], dtype=torch.float32)
        if out is not None:
            out[:] = r
            return out
        return r.to(x.device)

    @staticmethod
    def colorize(depth, vmin_vmax, lut_rgba, N, invalid_val, background_rgba, out, invalid_mask=None):
        import numpy as np
        from oracle import io_oracle
        v = depth.cpu().numpy().copy()
        if invalid_mask is not None:
            inv = invalid_mask.cpu().numpy().reshape(v.shape) != 0
        else:
            inv = (v == invalid_val) if invalid_val is not None else np.zeros(v.shape, bool)
        vmin, vmax = (np.float32(t) for t in vmin_vmax.tolist())
        v = (v - vmin) / (vmax - vmin) if vmin != vmax else v * np.float32(0)
        v[inv] = np.nan
        img = io_oracle.colormap_bytes(v, lut_rgba.cpu().numpy(), N)
        img[inv] = background_rgba
        out.view(depth.shape + (4,))[:] = torch.from_numpy(img)
        return out

    @staticmethod
    def depth_to_u16(depth, out, scale=256.0):
        out[:] = torch.from_numpy((depth.cpu().numpy() * depth.cpu().numpy().dtype.type(scale)).astype("uint16"))
        return out

    @staticmethod
    def silog_loss(pred, target, min_depth, max_depth, beta=0.15):
        mask = torch.logical_and(target > min_depth, target < max_depth)
        if int(mask.sum()) <= 1:
            return torch.zeros((), dtype=torch.float32, device=pred.device)
        g = torch.log(pred[mask] + 1e-7) - torch.log(target[mask] + 1e-7)
        return 10 * torch.sqrt(torch.var(g) + beta * torch.pow(torch.mean(g), 2))

    @staticmethod
    def depth_metrics(gt, pred, edges, min_depth, max_depth, crop, out13, additional_mask=None):
        import numpy as np
        from oracle import io_oracle
        if gt.shape != pred.shape:
            pred = F.interpolate(pred[None, None], tuple(gt.shape), mode="bilinear", align_corners=False)[0, 0]
        p = pred.cpu().numpy().copy()
        p[p < min_depth] = min_depth
        p[p > max_depth] = max_depth
        p[np.isinf(p)] = max_depth
        p[np.isnan(p)] = min_depth
        g = gt.cpu().numpy()
        m = np.logical_and(g > min_depth, g < max_depth)
        em = np.zeros(m.shape, bool)
        em[crop[0]:crop[1], crop[2]:crop[3]] = True
        m &= em
        if additional_mask is not None:
            m &= additional_mask.cpu().numpy() != 0
        gv, pv = g[m].astype(np.float32), p[m].astype(np.float32)
        th = np.maximum(gv / pv, pv / gv)
        lg, lp = np.log(gv), np.log(pv)
        f = np.float64
        s = [f(m.sum()), f((th < 1.25).sum()), f((th < 1.25 ** 2).sum()), f((th < 1.25 ** 3).sum()), (np.abs(gv - pv) / gv).astype(f).sum(),
             (((gv - pv) ** 2) / gv).astype(f).sum(), ((gv - pv) ** 2).astype(f).sum(), ((lg - lp) ** 2).astype(f).sum(),
             (lp - lg).astype(f).sum(), ((lp - lg) ** 2).astype(f).sum(), np.abs(np.log10(gv) - np.log10(pv)).astype(f).sum(), 0.0, 0.0]
        if edges is not None:
            me = np.logical_and(m, edges.cpu().numpy() != 0)
            if me.sum():
                s[11] = io_oracle.soft_edge_error(p, g)[me].astype(f).sum()
                s[12] = f(me.sum())
        out13[:] = torch.tensor(s, dtype=torch.float64)
        return out13


ops = FakeOps()
