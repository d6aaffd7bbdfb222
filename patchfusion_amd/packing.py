"""Weight packing for the gfx950 kernels (host side, runs once at load time).

Reference checkpoints store PyTorch layouts ([Cout,Cin,KH,KW] convs, [out,in] linears,
[Cin,Cout,k,k] transposed convs, BatchNorm running statistics).  The implicit-GEMM kernel
(csrc/igemm.hip) wants, per layer, a row-major [rows][Kpad] matrix in the compute dtype with
K ordered (ky, kx, c) over the channel layout of the NHWC *buffer* it will read (which may be a
concat buffer with padded / permuted channels), zero padded to the 128-byte K chunk, plus float
bias / per-channel scale vectors.  Everything here is exact re-arrangement except the BatchNorm
fold (eval-mode affine folded into weights and bias, guided_fusion_model.py:59-66).
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch


def round_up(x, m):
    return (x + m - 1) // m * m


@dataclass
class PackedConv:
    w: torch.Tensor                 # [rows, Kpad] compute dtype
    bias: Optional[torch.Tensor]    # float32 [>= cout] or None
    scale: Optional[torch.Tensor]   # float32 [>= cout] or None
    KH: int
    KW: int
    cin: int                        # valid input channels of the buffer (multiple of 8)
    cout: int                       # channels stored (multiple of 4); GEMM N
    cout_real: int
    shuffle: int = 1                # s for ConvTranspose2d(kernel=stride=s)
    korder: int = 0                 # K order of `w`: 0 = (ky, kx, c) tap-major; 1 = (c / 32, ky, kx, c % 32) chunk-major
    wino_m: int = 0                 # Winograd F(m x m, 3 x 3) output tile (2 | 4), 0 = direct kernel only
    wino_u: Optional[torch.Tensor] = None   # [(m+2)^2, rows, Kpad1] float32: G g G^T, each plane packed like a 1x1 weight
    wino_up: Optional[torch.Tensor] = None  # m = 4 only: the same filters in the fragment order of the fused kernel (winograd_filters_fused)
    wino_u3: Optional[torch.Tensor] = None  # m = 4 only: wino_u as the three bf16 planes of the split-precision GEMM, chunk-major [3, 36, Kpad1/32, rows, 32]
    w3: Optional[torch.Tensor] = None       # float32 1x1 layers with cin % 32 == 0: the weight's three bf16 planes, chunk-major [3, cin/32, rows, 32] (csrc/conv1x1_split3.hip)

    def to(self, device):
        self.w = self.w.to(device)
        if self.wino_u is not None:
            self.wino_u = self.wino_u.to(device)
        if self.wino_up is not None:
            self.wino_up = self.wino_up.to(device)
        if self.wino_u3 is not None:
            self.wino_u3 = self.wino_u3.to(device)
        if self.w3 is not None:
            self.w3 = self.w3.to(device)
        if self.bias is not None:
            self.bias = self.bias.to(device)
        if self.scale is not None:
            self.scale = self.scale.to(device)
        return self


def k_chunk(dtype):
    return 64 if dtype == torch.bfloat16 else 32


# Winograd transform matrices (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks"): F(2x2, 3x3) and F(4x4, 3x3) with
# the interpolation points {0, 1, -1} / {0, 1, -1, 2, -2}; B^T and A^T live in csrc/winograd.hip, G is applied here once per layer.
WINO_G = {
    2: [[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]],
    4: [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
}
WINO_BT = {
    2: [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
    4: [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
}
WINO_AT = {
    2: [[1, 1, 1, 0], [0, 1, -1, -1]],
    4: [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
}


def winograd_mode():
    """F(m x m, 3 x 3) tile of the float32 mode's large 3x3 layers: PF_WINOGRAD = 0 (direct kernels only) | 2 | 4 (default: 4x fewer
    multiplies; its float32 rounding, ~15x a direct convolution's per layer, does not show in the final depth -- DESIGN.md 4d)."""
    import os
    m = int(os.environ.get("PF_WINOGRAD", "4"))
    if m not in (0, 2, 4):
        raise ValueError("PF_WINOGRAD must be 0, 2 or 4")
    return m


def winograd_eligible(cout_store, cin_total, KH, KW, dtype):
    """layers worth the three-step path: float32, 3x3, whole 128-byte channel chunks on both sides, K = Cin of at least four chunks
    (the 8 / 32 / 64-channel 3x3 layers stay on the direct kernel)"""
    return dtype == torch.float32 and KH == 3 and KW == 3 and cin_total % 32 == 0 and cin_total >= 128 and cout_store % 32 == 0


def winograd_split3_enabled():
    """three-step Winograd layers: run the transform-domain GEMM in split precision (pf_conv_winograd_split3)?  PF_WINO_SPLIT3=0 disables
    (read at packing time: the three-plane filters are only built when enabled)"""
    import os
    return os.environ.get("PF_WINO_SPLIT3", "1") != "0"


def winograd_fused_only_eligible(cout_store, cin_total, KH, KW, dtype):
    """3x3 float32 layers BELOW the three-step form's channel threshold (32 <= Cin < 128, or Cout not a multiple of 32) that the fused
    kernel (csrc/wino_fused.hip) can still take: Cin a multiple of 16, Cout of 4.  They get the fused filters only; whether a call uses
    them is decided per call (hip_ops._fused_wanted: enough blocks), otherwise the direct kernel runs.  PF_WINO_FUSED_SMALL=0 disables."""
    import os
    if os.environ.get("PF_WINO_FUSED_SMALL", "1") == "0" or winograd_mode() != 4:
        return False
    return (dtype == torch.float32 and KH == 3 and KW == 3 and cin_total % 16 == 0 and cin_total >= 32 and cout_store % 4 == 0 and
            not winograd_eligible(cout_store, cin_total, KH, KW, dtype))


def winograd_applies(pc, pixels, stride, pad, act):
    """call-time half of the eligibility: the layer was packed with Winograd filters and this call is a 3x3 / stride 1 / pad 1
    convolution over at least PF_WINOGRAD_MIN_PIXELS pixels (default 0: measured 1.4 .. 4x the direct kernel on every eligible layer of the pass, from 8x392x518 down to 1x14x19, profiles/r2c_wino_tune.log) whose
    epilogue the output transform implements (bias, ReLU, residuals)."""
    if (pc.wino_u is None and pc.wino_up is None) or stride != 1 or pad != 1 or act not in (None, "none", "relu"):
        return False
    import os
    return pixels >= int(os.environ.get("PF_WINOGRAD_MIN_PIXELS", "0"))


def winograd_filters(wk, m):
    """wk [rows, 3, 3, cin_total] float32 (buffer channel layout) -> U [(m+2)^2, rows, Kpad] float32, U[i*(m+2)+j] = (G g G^T)[i][j];
    evaluated in float64, rounded once."""
    G = torch.tensor(WINO_G[m], dtype=torch.float64)
    rows, _, _, cin_total = wk.shape
    U = torch.einsum("ia,raxc,jx->ijrc", G, wk.double(), G)            # [A, A, rows, cin]
    A = m + 2
    Kpad = round_up(cin_total, 32)
    out = torch.zeros(A * A, rows, Kpad, dtype=torch.float32)
    out[:, :, :cin_total] = U.reshape(A * A, rows, cin_total).float()
    return out.contiguous()


def winograd_filters_fused(wk):
    """Filters of the fused F(4x4,3x3) kernel (csrc/wino_fused.hip): wk [rows, 3, 3, cin_total] float32 (buffer channel layout,
    cin_total % 8 == 0) -> [nnb, nkc, 36, 2, 64, 4] float32 in MFMA FRAGMENT order: for the 64-channel block nb, the 8-channel chunk
    kc, the transform point `plane` and the channel half, lane (r = lane & 15, g = lane >> 4) holds
        (U[n0 + 0 + r][c0], U[n0 + 0 + r][c0 + 1], U[n0 + 16 + r][c0], U[n0 + 16 + r][c0 + 1]),   n0 = 64 nb + 32 half, c0 = 8 kc + 2 g
    i.e. the A operands of the two k-steps of v_mfma_f32_16x16x4_f32 for the wave's two 16-channel groups: one contiguous 1 KiB per
    wave-load.  U = G g G^T in float64, rounded once; channels beyond `rows` are zero."""
    G = torch.tensor(WINO_G[4], dtype=torch.float64)
    rows, _, _, cin = wk.shape
    assert cin % 8 == 0
    nnb, nkc = (rows + 63) // 64, cin // 8
    U = torch.einsum("ia,raxc,jx->ijrc", G, wk.double(), G).reshape(36, rows, cin).float()
    Up = torch.zeros(36, nnb * 64, cin, dtype=torch.float32)
    Up[:, :rows] = U
    # dims: plane, nb, half, cg, r, kc, g, ks  ->  nb, kc, plane, half, g, r, cg, ks   (lane = 16 g + r, element = 2 cg + ks)
    Up = Up.view(36, nnb, 2, 2, 16, nkc, 4, 2).permute(1, 5, 0, 2, 6, 4, 3, 7).contiguous()
    return Up.view(nnb, nkc, 36, 2, 64, 4)


def pack_conv(weight, bias=None, *, dtype, cin_map=None, cin_total=None, scale=None, bn=None, bn_eps=1e-5):
    """weight [Cout, Cin, KH, KW] (or [Cout, Cin] for nn.Linear).  ``cin_map``: list of
    (src_start, length, dst_start) placing original input channels into the buffer's channel axis;
    ``cin_total``: number of valid buffer channels (multiple of 8)."""
    w = weight.detach().float().cpu()
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, KH, KW = w.shape
    b = None if bias is None else bias.detach().float().cpu().clone()
    if bn is not None:                                       # fold eval-mode BatchNorm
        g, beta, mean, var = [t.detach().float().cpu() for t in bn]
        f = g / torch.sqrt(var + bn_eps)
        w = w * f.view(-1, 1, 1, 1)
        b = (beta - mean * f) if b is None else (b - mean) * f + beta
    if cin_map is None:
        cin_map = [(0, cin, 0)]
    if cin_total is None:
        cin_total = round_up(max(d + n for _, n, d in cin_map), 8)
    assert cin_total % 8 == 0
    cout_store = round_up(cout, 4)
    rows = round_up(cout_store, 16)
    wk = torch.zeros(rows, KH, KW, cin_total)
    for s0, n, d0 in cin_map:
        wk[:cout, :, :, d0:d0 + n] = w[:, s0:s0 + n].permute(0, 2, 3, 1)
    K = KH * KW * cin_total
    Kpad = round_up(K, k_chunk(dtype))
    wp = torch.zeros(rows, Kpad)
    # float32 k x k layers whose channels fill whole 128-byte chunks: CHUNK-MAJOR K order -- the kernel then walks all
    # KH*KW taps of one 32-channel chunk back to back, so the 9 re-reads of an input pixel hit L2 (the f32 activation
    # tensors are GBs: with the tap-major order every tap pass streams them from HBM again, 64 GB per dominant launch)
    korder = 1 if (dtype == torch.float32 and KH * KW > 1 and cin_total % 32 == 0) else 0
    if korder:
        wp[:, :K] = wk.reshape(rows, KH * KW, cin_total // 32, 32).permute(0, 2, 1, 3).reshape(rows, K)
    else:
        wp[:, :K] = wk.reshape(rows, K)
    bp = None
    if b is not None:
        bp = torch.zeros(rows)
        bp[:cout] = b
    sp = None
    if scale is not None:
        sp = torch.zeros(rows)
        sp[:cout] = scale.detach().float().cpu()
    wm = winograd_mode() if (scale is None and winograd_eligible(cout_store, cin_total, KH, KW, dtype)) else 0
    wu = winograd_filters(wk, wm) if wm else None
    wup = winograd_filters_fused(wk) if wm == 4 else None
    # the three bf16 planes of U for the split-precision GEMM, CHUNK-MAJOR [3, 36, K/32, rows, 32] (csrc/gemm_split3.hip korder bit 2): every
    # 32-channel chunk of all filter rows is one contiguous slab
    wu3 = (torch.stack(split3(wu)).view(3, wu.shape[0], wu.shape[1], wu.shape[2] // 32, 32).permute(0, 1, 3, 2, 4).contiguous()
           if (wm == 4 and winograd_split3_enabled()) else None)
    if wm == 0 and scale is None and winograd_fused_only_eligible(cout_store, cin_total, KH, KW, dtype):
        wm, wup = 4, winograd_filters_fused(wk)                   # fused kernel only (wino_u stays None: no three-step form)
    # float32 1x1 layers: the weight again as three bf16 planes for the split-precision 1x1 kernel that splits its float32 activations itself
    # (csrc/conv1x1_split3.hip; hip_ops routes by token count).  1.5x the float32 weight's bytes; these layers are small.
    w3 = (rows_to_kmajor(torch.stack(split3(wk.reshape(rows, cin_total))).contiguous())
          if (dtype == torch.float32 and KH == 1 and KW == 1 and cin_total % 32 == 0 and conv1x1_split3_enabled()) else None)
    return PackedConv(wp.to(dtype).contiguous(), bp, sp, KH, KW, cin_total, cout_store, cout, korder=korder, wino_m=wm, wino_u=wu, wino_up=wup, wino_u3=wu3,
                      w3=w3)


def conv1x1_split3_enabled():
    """float32 1x1 convolutions / linears through the split-precision kernel with the in-loader split (csrc/conv1x1_split3.hip)?  PF_CONV1X1_SPLIT3=0 / 1,
    default on (read at pack time: the planes are only packed when it is on)"""
    import os
    return os.environ.get("PF_CONV1X1_SPLIT3", "1") != "0"


def split3(x):
    """float32 tensor -> three bfloat16 tensors (h, m, l) with h + m + l == x exactly for 1e-30 < |x| < 3.39e38 = the bf16 maximum (round-to-nearest splits; beyond that range the leading plane overflows / the third plane underflows)"""
    x = x.float()
    h = x.to(torch.bfloat16)
    r = x - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return h, m, l


def split3_kmajor_enabled():
    """operands of the split-precision linears in the chunk-major layout [K/32][rows][32]?  PF_SPLIT3_KMAJOR=0 / 1, default on"""
    import os
    return os.environ.get("PF_SPLIT3_KMAJOR", "1") != "0"


def rows_to_kmajor(t):
    """[..., rows, K] -> chunk-major [..., K/32, rows, 32] (contiguous)"""
    *lead, rows, K = t.shape
    assert K % 32 == 0
    n = len(lead)
    return t.reshape(*lead, rows, K // 32, 32).permute(*range(n), n + 1, n, n + 2).contiguous()


def kmajor_to_rows(t):
    """chunk-major [..., K/32, rows, 32] -> [..., rows, K]"""
    *lead, nk, rows, c = t.shape
    n = len(lead)
    return t.permute(*range(n), n + 1, n, n + 2).reshape(*lead, rows, nk * c)


def pack_conv_split3(weight, bias=None, *, scale=None, kmajor=None):
    """nn.Linear / 1x1 conv weight [Cout, Cin] for the split-precision GEMM (csrc/gemm_split3.hip): PackedConv with w = [3, rows, Kpad]
    bfloat16 planes (h, m, l), K padded to 32, rows to 16; bias / scale float32 as in pack_conv.  kmajor (default: split3_kmajor_enabled()):
    the planes are CHUNK-MAJOR, w = [3, Kpad/32, rows, 32] -- every 32-deep K chunk of all rows is one contiguous slab (csrc/gemm_split3.hip)."""
    w = weight.detach().float().cpu()
    if w.dim() == 4:
        assert w.shape[2] == 1 and w.shape[3] == 1
        w = w[:, :, 0, 0]
    cout, cin = w.shape
    assert cin % 32 == 0, "split GEMM: K must be a multiple of 32"
    cout_store = round_up(cout, 4)
    rows = round_up(cout_store, 16)
    wp = torch.zeros(rows, cin)
    wp[:cout] = w
    planes = torch.stack(split3(wp)).contiguous()
    if split3_kmajor_enabled() if kmajor is None else kmajor:
        planes = rows_to_kmajor(planes)
    bp = None
    if bias is not None:
        bp = torch.zeros(rows)
        bp[:cout] = bias.detach().float().cpu()
    sp = None
    if scale is not None:
        sp = torch.zeros(rows)
        sp[:cout] = scale.detach().float().cpu()
    return PackedConv(planes, bp, sp, 1, 1, cin, cout_store, cout)


def pack_conv_transpose(weight, bias, *, dtype):
    """nn.ConvTranspose2d(kernel=stride=s, padding=0) (dpt.py:41-52): weight [Cin, Cout, s, s].
    out[b, y*s+dy, x*s+dx, co] = bias[co] + sum_ci x[b,y,x,ci] * weight[ci,co,dy,dx]  -> a GEMM with
    N = s*s*Cout rows ordered (dy, dx, co) and a pixel-shuffle store."""
    w = weight.detach().float().cpu()
    cin, cout, s, s2 = w.shape
    assert s == s2 and cout % 4 == 0 and cin % 8 == 0
    rows_real = s * s * cout
    rows = round_up(rows_real, 16)
    Kpad = round_up(cin, k_chunk(dtype))
    wp = torch.zeros(rows, Kpad)
    wp[:rows_real, :cin] = w.permute(2, 3, 1, 0).reshape(rows_real, cin)
    bp = torch.zeros(round_up(cout, 16))
    bp[:cout] = bias.detach().float().cpu()
    return PackedConv(wp.to(dtype).contiguous(), bp, None, 1, 1, cin, rows_real, rows_real, shuffle=s)


def unpack_conv(pc: PackedConv):
    """Inverse of pack_conv for test back-ends: returns weight [cout, cin, KH, KW] float32 over the
    buffer's channel axis (tests/fake_ops.py uses it to drive F.conv2d with the PACKED weights, so the
    packing itself is covered by the CPU wiring tests)."""
    K = pc.KH * pc.KW * pc.cin
    w = pc.w.float()[:pc.cout, :K]
    if pc.korder:
        w = w.reshape(pc.cout, pc.cin // 32, pc.KH * pc.KW, 32).permute(0, 2, 1, 3)
    return w.reshape(pc.cout, pc.KH, pc.KW, pc.cin).permute(0, 3, 1, 2).contiguous()


@dataclass
class BinsTail:
    """Weights of the fused metric-bins tail kernel (csrc/imageops.hip bins_tail_kernel): the two 1x1 layers of the conditional log-binomial
    MLP (zoedepth_v1.py:207-213, dist_layers.py:97-110) in the kernel's order.  `mlp0` / `mlp2` are kept for the test back-end."""
    w0f: torch.Tensor               # [nq, 5, 64, 4] float32: lane (r = l & 15, g = l >> 4) of fragment f, K group q: W0[16 f + r][16 q + 4 g + e]
    b0: torch.Tensor                # [80]
    w2: torch.Tensor                # [4, 80]
    b2: torch.Tensor                # [4]
    nq: int                         # K groups of 16 CLB-buffer channels (10: last + embedding; 11: + rel)
    rel_off: int                    # channel offset of rel in the CLB buffer (nq == 11)
    mlp0: PackedConv
    mlp2: PackedConv

    def to(self, device):
        self.w0f, self.b0, self.w2, self.b2 = (t.to(device) for t in (self.w0f, self.b0, self.w2, self.b2))
        return self


def bins_tail_weights(mlp0: PackedConv, mlp2: PackedConv, emb: int):
    """-> BinsTail, or None when the head is not the shape the fused kernel is written for (float32, 32 `last` channels + 128 embedding
    channels (+ 8 rel) -> 80 hidden -> 4)."""
    if mlp0.w.dtype != torch.float32 or mlp0.KH != 1 or mlp2.KH != 1 or emb != 128 or mlp0.cout_real != 80 or mlp2.cout_real != 4:
        return None
    if mlp0.cin not in (160, 168) or mlp2.cin != 80 or mlp0.bias is None or mlp2.bias is None or mlp0.scale is not None or mlp2.scale is not None:
        return None
    W0 = unpack_conv(mlp0).cpu()[:80, :, 0, 0]                      # [80, ctot] over the CLB buffer's channel order
    ctot = W0.shape[1]
    nq = (ctot + 15) // 16
    Wp = torch.zeros(80, nq * 16)
    Wp[:, :ctot] = W0
    w0f = Wp.view(5, 16, nq, 4, 4).permute(2, 0, 3, 1, 4).reshape(nq, 5, 64, 4).contiguous()       # (f, r, q, g, e) -> (q, f, g, r, e)
    W2 = unpack_conv(mlp2).cpu()[:4, :80, 0, 0].contiguous()
    return BinsTail(w0f, mlp0.bias.cpu()[:80].clone().float(), W2, mlp2.bias.cpu()[:4].clone().float(), nq, 160, mlp0, mlp2)


def vit_pos_embed(pos_embed, th, tw):
    """interpolate_pos_encoding (vision_transformer.py:179-210): bicubic resample of the stored
    37x37 grid to (th, tw) with the +0.1 offset passed through ``scale_factor``.  It depends only on
    the weights and the (fixed) process shape, so it is evaluated ONCE on the host at load time
    instead of in every forward like the reference does."""
    import torch.nn.functional as F
    pos = pos_embed.detach().float().cpu()
    n = pos.shape[1] - 1
    g = int(math.sqrt(n))
    if th * tw == n and th == tw:
        return pos[0].contiguous()
    dim = pos.shape[-1]
    grid = pos[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(float(th + 0.1) / g, float(tw + 0.1) / g), mode="bicubic", antialias=False)
    assert grid.shape[-2] == th and grid.shape[-1] == tw
    grid = grid.permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat([pos[0, :1], grid], dim=0).contiguous()           # [1 + th*tw, D]
