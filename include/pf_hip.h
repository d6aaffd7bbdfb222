/* C ABI of libpf_hip.so -- the MI355X (gfx950) kernels behind PatchFusion's tiled-inference hot path.
 *
 * Plain pointers and sizes only (no torch types).  Every entry point launches asynchronously on
 * the hipStream_t passed as `stream` (a void*), borrows the device pointers for the duration of
 * the call and returns an int status (PF_OK = 0; message via pf_last_error()).  The Python host
 * (patchfusion_amd/hip_ops.py) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.  Each group cites the reference interface (file:line under the reference
 * repo) whose stock PyTorch / torchvision / cuDNN op it replaces.
 *
 * dtype: 0 = float32 activations + f32-input MFMA ("exact"), 1 = bf16 activations + bf16 MFMA with
 * f32 accumulation ("fast").  All activation tensors are NHWC with an explicit pixel stride `ld`
 * (elements) so producers write directly into channel slices of concat buffers.
 */
#ifndef PF_HIP_H
#define PF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_DTYPE_F32 0
#define PF_DTYPE_BF16 1

#define PF_ACT_NONE 0
#define PF_ACT_RELU 1
#define PF_ACT_GELU 2      /* exact erf GELU (nn.GELU default) */
#define PF_ACT_SOFTPLUS 3  /* nn.Softplus(beta=1, threshold=20) */

const char* pf_last_error(void);
int pf_version(void);

/* ---- implicit-GEMM convolution / linear layer on the matrix cores --------------------------
 * y[b,oy,ox,n] = epi( sum_{ky,kx,c} x[b, oy*stride-pad+ky, ox*stride-pad+kx, c] * w[n,ky,kx,c] )
 * epi(v) = (act(v + bias[n]) * scale[n]) + res[...] + res2[...]
 * Replaces: F.conv2d / nn.Linear / nn.ConvTranspose2d(k==s) + fused bias/ReLU/GELU/LayerScale/
 * residual everywhere on the path: dinov2/layers/attention.py:51,60, mlp.py:35-41, block.py:82-107,
 * patch_embed.py:76, depth_anything/dpt.py:30-63,87-95, blocks.py:53-59,117,
 * zoedepth layers localbins_layers.py:87-92,112-116, attractor.py:156-161, dist_layers.py:88-95,
 * estimator/models/patchfusion.py:121-127, blocks/guided_fusion_model.py:41-48,59-66,
 * blocks/swin_layers.py:39-41,125-127.
 * Weights are pre-packed by the host as [w_rows][Kpad], zero padded, in `dtype`; K order = (ky,kx,c) (`korder` 0) or, for
 * float32 k x k layers with Cin % 32 == 0, chunk-major (c/32, ky, kx, c%32) (`korder` 1: all taps of one 32-channel chunk
 * are adjacent, so the re-reads of an input pixel hit L2) -- patchfusion_amd/packing.py.
 */
typedef struct {
  const void* x; int x_ld; int B, H, W, Cin; /* Cin: valid input channels, multiple of 8 (zero weights on pad) */
  const void* w; int w_rows; int Kpad;      /* Kpad multiple of 64 (bf16) / 32 (f32) elements */
  const float* bias; const float* scale;    /* [Cout] or NULL (indexed by output channel) */
  const void* res; int res_ld; const void* res2; int res2_ld; /* residual(s), NHWC like y, or NULL */
  void* y; int y_ld; int OH, OW, Cout;      /* Cout: GEMM N = channels stored, multiple of 4 */
  int KH, KW, stride, pad;
  int act;       /* PF_ACT_* */
  int relu_in;   /* apply ReLU to x on load (ResidualConvUnit, blocks.py:78,83) */
  int out_f32;   /* store float32 even when dtype == bf16 (bins-head tensors) */
  int shuffle;   /* s>1: ConvTranspose2d(kernel=stride=s): N = s*s*Cout_t, y is [B,OH*s,OW*s,Cout_t] */
  int dtype;
  int korder;    /* K order of w: 0 = (ky,kx,c); 1 = (c/32, ky, kx, c%32), f32 only, Cin % 32 == 0 (see packing.py) */
  int batch;     /* > 1: that many independent GEMM planes in ONE launch (float32, no bias / scale / residual): plane k reads
                    x + k*x_bstride, w + k*w_bstride and writes y + k*y_bstride (strides in elements); 0 / 1 = a single plane.
                    Used for the (m+2)^2 transform points of a Winograd layer (pf_conv_winograd). */
  long x_bstride, w_bstride, y_bstride;
} pf_conv_params;
int pf_conv(const pf_conv_params* p, void* stream);
/* timing helper for the roofline entry of bench.py: runs `iters` launches bracketed by HIP events on
 * `stream`, returns the average milliseconds per launch in *ms */
int pf_conv_timed(const pf_conv_params* p, int iters, float* ms, void* stream);

/* Winograd F(m x m, 3 x 3), m = 2 or 4, for float32 3x3 / stride 1 / pad 1 layers (same reference layers as pf_conv): `p` describes
 * the convolution exactly as for pf_conv (x, bias, res, res2, y, act NONE | RELU, relu_in; p->w is not read; scale must be NULL,
 * Cin % 32 == 0, Cout % 8 == 0); U = the (m+2)^2 transformed filters G g G^T, each packed like a 1x1 pf_conv weight [u_rows][u_kpad]
 * (patchfusion_amd/packing.py winograd_filters); V, M = device workspaces of (m+2)^2 * T * Cin and (m+2)^2 * T * Cout floats,
 * T = B * ceil(H/m) * ceil(W/m).  Input transform -> (m+2)^2 GEMMs through pf_conv -> output transform + epilogue (winograd.hip).
 * m = 2 multiplies 2.25x less than the direct convolution at ~2.5x its float32 rounding error, m = 4 4x less at ~15x. */
int pf_conv_winograd(const pf_conv_params* p, int m, const void* U, int u_rows, int u_kpad, void* V, void* M, void* stream);
/* The same three steps with the transform-domain GEMM in split precision (m = 4): V3 = workspace of 3 x 36 x T x Cin bf16 (the input transform
 * writes the three planes, chunk-major: [3][36][Cin/32][T][32]), U3 = the three bf16 planes of G g G^T, chunk-major [3][36][Cin/32][u_rows][32]
 * (patchfusion_amd/packing.py PackedConv.wino_u3; u_kpad must equal Cin), M = 36 x T x Cout float32; one batched pf_gemm_split3 launch
 * (korder = 6) between the transforms. */
int pf_conv_winograd_split3(const pf_conv_params* p, const void* U3, int u_rows, int u_kpad, void* V3, void* M, void* stream);
/* the same layer `window` Winograd tiles at a time (window % 8 == 0; 0 = all tiles at once): V3 / M then are arenas for ONE window (3 x 36 x window x Cin
 * bf16, 36 x window x Cout float32) that every window reuses -- small windows keep the pair inside the 256 MB memory-side cache.  Tiles are independent:
 * identical results for every window.  With more than one window the output of window k is written before the input of window k + 1 (and its one-pixel
 * halo) is read: x and y must NOT overlap then (PF_ERR_ARG); a residual may alias y (it is read at the pixels being written). */
int pf_conv_winograd_split3_windowed(const pf_conv_params* p, const void* U3, int u_rows, int u_kpad, void* V3, void* M, long window, void* stream);

/* FUSED Winograd F(4x4, 3x3) (csrc/wino_fused.hip): the same layers in ONE kernel -- the transformed input and the transform-domain
 * products never exist in HBM.  `p` as for pf_conv_winograd (p->w is not read); `up` = the filters in MFMA fragment order
 * [nnb][Cin/8][36][2][64][4] floats (patchfusion_amd/packing.py winograd_filters_fused), nnb = ceil(Cout/64); `gs` & 0xffff =
 * super-tiles (32 output tiles: 4 x 8 or 8 x 4, picked for the least padding) per block group (L2 locality knob, >= 1), `gs` >> 16 = 0
 * or a forced super-tile width 8 | 4 (tuning aid).  Needs Cin % 16 == 0, Cin >= 32, Cout % 4 == 0, B*H*W*x_ld < 2^31:
 * pf_conv_winograd_fused_supported(p) returns 1 when `p` qualifies, else 0 (callers then take pf_conv_winograd / pf_conv). */
int pf_conv_winograd_fused_supported(const pf_conv_params* p);
int pf_conv_winograd_fused(const pf_conv_params* p, const void* up, int nnb, int gs, void* stream);
/* timing helper like pf_conv_timed: `iters` launches bracketed by HIP events on `stream` */
int pf_conv_winograd_fused_timed(const pf_conv_params* p, const void* up, int nnb, int gs, int iters, float* ms, void* stream);

/* ---- split-precision linear layer (csrc/gemm_split3.hip; the float32 mode's ViT block linears) --------------------------------------------
 * float32-grade y = epi(x . w^T) on the bf16 matrix cores: x and w are given as THREE bf16 planes each (x = x_h + x_m + x_l, round-to-
 * nearest splits) and the six leading partial products are accumulated in float32.  `p` as for pf_conv with KH = KW = 1: x = planes
 * [3][M][x_ld] bf16 (plane stride x_bstride elements), w = planes [3][w_rows][Kpad] bf16 (w_bstride; packing.pack_conv_split3), Cin % 32
 * == 0, bias / scale / res / res2 float32; y = float32 [M][y_ld] when out_f32 != 0, else three bf16 planes [3][M][y_ld] (y_bstride) for a
 * following split GEMM.  Same reference layers as pf_conv's linear use (attention.py:51,60, mlp.py:35-41).
 * p->korder here selects the OPERAND layout: bit 1 (value 2) = every x plane is chunk-major [Cin/32][M][32] (x_ld must equal Cin), bit 2
 * (value 4) = every w plane is chunk-major [Cin/32][w_rows][32] (Kpad must equal Cin): each 32-deep K chunk of all rows is one contiguous
 * slab, which is what the kernel's 1-KiB LDS-DMA pieces want (whole cache lines); bit 3 (value 8) = the three-plane OUTPUT (out_f32 == 0) is
 * written chunk-major [Cout/32][M][32] (y_ld must equal Cout, Cout % 32 == 0) for a following split GEMM.  p->batch > 1: that many independent planes in one launch
 * (block k of every x / w plane, float32 output block k, no epilogue) -- the transform points of pf_conv_winograd_split3.
 * Kernel choice is internal and result-identical (same chunk order, same six terms per accumulator): 64 x 128 tiles below one round of tiles,
 * one 128 x 128 tile per block, or -- from two rounds of tiles on -- a PERSISTENT kernel (one block per CU walks its tiles) with 128 x 128 tiles
 * on a three-slot LDS ring or 192 x 192 tiles on a two-slot ring, whichever costs fewer rounds x tile cost. */
int pf_gemm_split3(const pf_conv_params* p, void* stream);
/* The same split-precision product for a 1x1 convolution / linear layer whose ACTIVATIONS are plain float32 (round 6; replaces pf_conv's f32-MFMA
 * route for the 1x1 layers of the DPT / metric-bins heads -- external/zoedepth/models/layers/localbins_layers.py:99-117, attractor.py:156-161,
 * base_models/dpt_dinov2/blocks.py (out_conv / projections) -- and the G2L Swin linears, estimator/models/blocks/swin_layers.py:120-128,133-164):
 * `p` exactly as for pf_conv with KH = KW = 1, stride 1, pad 0, shuffle 1, dtype PF_DTYPE_F32 (x float32 [M][x_ld], x_ld % 4 == 0, Cin % 32 == 0;
 * bias / scale / res / res2 / y float32; act, relu_in as pf_conv; p->w is not read); w3 = the weight's three bf16 planes, CHUNK-MAJOR
 * [3][Cin/32][w3_rows][32] (packing.pack_conv_split3(kmajor=True); w3_rows % 16 == 0, >= Cout).  The kernel splits x in its loader (three LDS
 * planes per K chunk), so no producer has to change; error against float64 = pf_gemm_split3's (tests/op_checks.py conv1x1_split3). */
int pf_conv1x1_split3(const pf_conv_params* p, const void* w3, int w3_rows, void* stream);
/* the same call with the persistent kernels capped at `grid_cap` blocks (0 = one per CU): the CUs left over run the kernels of OTHER streams
 * (the HBM-bound Winograd transforms of the other tile batch) beside the matrix-bound GEMM */
int pf_gemm_split3_ex(const pf_conv_params* p, int grid_cap, void* stream);
/* which of those kernels a call would run on a chip of `cus` compute units (no launch, no GPU needed: the dispatch rule for host-side tests) */
#define PF_S3_ROUTE_TILE64 0
#define PF_S3_ROUTE_TILE128 1
#define PF_S3_ROUTE_PERSIST128 2
#define PF_S3_ROUTE_PERSIST192 3
int pf_gemm_split3_route(const pf_conv_params* p, int cus);
int pf_gemm_split3_timed(const pf_conv_params* p, int iters, float* ms, void* stream);
/* A plain bf16 linear layer (x [M][x_ld] bf16, w from packing.pack_conv, bf16 residuals / output, float32 output when out_f32) through the same
 * ping-pong LDS-DMA pipeline: 256 x 128 tiles, Cin % 64 == 0.  The bf16 mode's ViT block linears at large token counts (same reference layers). */
int pf_gemm_bf16_pp(const pf_conv_params* p, void* stream);
/* the split producers of the ViT block: LayerNorm (layers/block.py:88-93 norm1 / norm2) and the attention output (attention.py:58-60)
 * written as three bf16 planes; arguments as pf_layernorm (plain row range) / pf_vit_attention_qkv with the plane stride in elements.
 * kmajor != 0: every output plane is CHUNK-MAJOR [cols/32][rows][32] (rows = all rows of the call), the layout pf_gemm_split3 reads with
 * korder bit 1; kmajor == 0: row-major [rows][ld]. */
int pf_layernorm_split3(const float* x, int x_ld, void* y3, int y_ld, long plane, int kmajor, const float* g, const float* b, float eps,
                        long rows, int D, void* stream);
int pf_vit_attention_qkv_split3(const void* qkv, void* out3, long plane, int kmajor, int B, int S, int Hh, void* stream);
/* The same attention entirely in split precision: qkv3 = the QKV GEMM's output as three bf16 planes [3][B*S][3*Hh*64] (plane stride plane_in
 * elements), out3 = three bf16 planes [3][B*S][Hh*64] (plane_out); S^T = K.Q^T and O^T = V^T.P^T as six bf16 partial products each with float32
 * accumulation, float32 softmax with the probabilities split in registers (csrc/vit.hip vit_attention_split3_kernel; attention.py:53-60). */
int pf_vit_attention_split3(const void* qkv3, long plane_in, void* out3, long plane_out, int kmajor, int B, int S, int Hh, void* stream);
/* Version 2 of the same operator (csrc/attn_split3.hip; same operands): K / V tiles by LDS-DMA, V^T fragments by the transposing LDS read, 32 (or
 * 16) queries per wave.  queries_per_wave: 16, 32, or 0 = 32 when the launch fills the chip with 128-query blocks, else 16.  schedule: 1 = the
 * two-phase kernel (64-key tiles; bit-identical to pf_vit_attention_split3), 2 = the software-pipelined kernel (32-key blocks, QK / softmax / PV of
 * three consecutive blocks overlapped inside every wave; float32-rounding-identical), 0 = default (2).  Requires S * Hh * 384 bytes < 2^31 (32-bit
 * row offsets inside one image).  Replaces dinov2/layers/attention.py:53-60. */
int pf_vit_attention_split3_v2(const void* qkv3, long plane_in, void* out3, long plane_out, int kmajor, int B, int S, int Hh, int queries_per_wave,
                               int schedule, void* stream);
/* float32 [rows][x_ld] -> three bf16 planes [3][rows][y_ld], plane stride `plane` elements (the split producers fuse into their stores) */
int pf_split3(const float* x, int x_ld, void* y, int y_ld, long plane, long rows, int cols, void* stream);

/* ---- ViT encoder pieces ---------------------------------------------------------------------- */
/* (x - mean)/std + 14x14/14 patch gather: NCHW float image -> im2col rows [B*th*tw][ld] (K order
 * ky,kx,c).  Replaces depth_anything.py:184-190 (Normalize) + the unfold implied by patch_embed.py:76 */
int pf_patch_im2col(const float* img, int B, int H, int W, void* out, int ld, int dtype, void* stream);
/* tokens[b,0,:] = cls + pos[0]; tokens[b,1+t,:] = emb[b*(S-1)+t,:] + pos[1+t]  (vision_transformer.py:222-223) */
int pf_assemble_tokens(const void* emb, void* tokens, const float* cls, const float* pos, int B, int S, int D, int dtype,
                       void* stream);
/* LayerNorm over the last dim: y[r] = (x[r]-mu)/sqrt(var+eps)*g+b.  Rows are remapped so that the
 * final `norm` + cls drop of get_intermediate_layers (vision_transformer.py:309-312) is one pass:
 * out row (b, t) <- in row b*in_rows_per_batch + in_row_offset + t, t < out_rows_per_batch. */
int pf_layernorm(const void* x, int x_ld, void* y, int y_ld, const float* g, const float* b, float eps,
                 int batches, int in_rows_per_batch, int in_row_offset, int out_rows_per_batch, int D,
                 int dtype, void* stream);
/* split the fused qkv rows [B*S][3*D] into per-head Q*scale [B,H,S,64], K [B,H,S,64], V^T [B,H,64,Sp] */
int pf_qkv_split(const void* qkv, int B, int S, int Hh, void* q, void* k, void* vt, int Sp, float scale,
                 int dtype, void* stream);
/* softmax(q k^T) v per (batch, head), head_dim 64, flash-style on the matrix cores; out [B*S][D].
 * Replaces dinov2/layers/attention.py:53-59 (the materialised N x N scores). */
int pf_vit_attention(const void* q, const void* k, const void* vt, void* out, int B, int S, int Sp, int Hh,
                     int dtype, void* stream);
/* float32 attention straight from the QKV GEMM's rows [B*S][3][Hh][64] (no pf_qkv_split, no Q / K / V^T buffers): softmax(q k^T / 8) v
 * per head, head_dim 64, out [B*S][Hh*64].  Same reference lines as pf_vit_attention (dinov2/layers/attention.py:49-62); base-2 softmax
 * with the hardware exponential (relative error ~1e-7 per probability).  dtype must be PF_DTYPE_F32. */
int pf_vit_attention_qkv(const void* qkv, void* out, int B, int S, int Hh, int dtype, void* stream);

/* ---- G2L (Swin window attention) -------------------------------------------------------------- */
/* LayerNorm(norm1) -> zero pad to a multiple of 12 -> cyclic shift -> window partition
 * (swin_layers.py:223-244): x [B,H,W,C] tokens -> xw [B*nW*144][C] */
int pf_swin_ln_partition(const void* x, int x_ld, void* xw, const float* g, const float* b, float eps,
                         int B, int H, int W, int C, int shift, int dtype, void* stream);
/* window attention with relative-position bias and shift mask (swin_layers.py:133-164,327-345);
 * qkv [B*nW*144][3C] -> out [B*nW*144][C]; bias_table [529][heads] float */
int pf_swin_window_attention(const void* qkv, void* out, const float* bias_table, int B, int Hp, int Wp,
                             int C, int heads, int shift, int dtype, void* stream);
/* window reverse + un-shift + crop + residual (swin_layers.py:250-263): y = shortcut + proj[win(tok)] */
int pf_swin_unpartition_add(const void* proj, const void* shortcut, int s_ld, void* y, int y_ld, int B, int H,
                            int W, int C, int shift, int dtype, void* stream);
/* x[b,t,:] += pos[t,:]  (absolute_pos_embed, swin_layers.py:419-420); pos float [T][C] */
int pf_add_rowwise(void* x, int x_ld, const float* pos, int B, int T, int C, int dtype, void* stream);

/* ---- memory-bound image ops --------------------------------------------------------------------- */
/* bilinear, align_corners=True (F.interpolate; used at dpt.py:126,154, blocks.py:147,
 * attractor.py:179,186, zoedepth_v1.py:204-214, guided_fusion_model.py:97,192).
 * y[b,oy,ox, 0..C) = (add ? add[...] : 0) + interp(x).  NHWC both sides. */
int pf_resize_bilinear(const void* x, int x_ld, int B, int H, int W, int C, void* y, int y_ld, int OH, int OW,
                       const void* add, int add_ld, int in_f32, int out_f32, int dtype, void* stream);
/* nsrc (2 or 3) bilinear align_corners=True resizes of NHWC tensors xs[i] [B,Hs[i],Ws[i],Cs[i]] (pixel stride lds[i]) written to
 * consecutive channel ranges of y [B,OH,OW, sum Cs] (pixel stride y_ld): the Upv1 concat of guided_fusion_model.py:96-99
 * (cat[feat_enc_resized, up(temp), up(guide)]) in one launch.  All tensors in `dtype`.  The five arrays are HOST arrays. */
int pf_resize_concat(const void* const* xs, const int* lds, const int* Hs, const int* Ws, const int* Cs, int nsrc, int B,
                     void* y, int y_ld, int OH, int OW, int dtype, void* stream);
/* planar float version for the image crops (depth_anything/transform.py:127-129 applied per tile,
 * baseline_pretrain.py:258-264): img [3][H][W] float -> out [P][3][oh][ow] float; boxes int [P][4]=(x0,y0,x1,y1) */
int pf_crop_resize_planar(const float* img, int C, int H, int W, const int* boxes, int P, float* out, int oh, int ow,
                          void* stream);
/* torchvision.ops.roi_align(aligned=True, sampling_ratio=-1) (patchfusion.py:247,251;
 * guided_fusion_model.py:202). feat [Bf,H,W,C] NHWC; rois float [K][5] = (batch, x1,y1,x2,y2);
 * out [K,oh,ow,C] (ld y_ld).  in_f32/out_f32 select float storage for the depth map. */
int pf_roi_align(const void* feat, int f_ld, int Bf, int H, int W, int C, const float* rois, int K, void* y,
                 int y_ld, int oh, int ow, float spatial_scale, int in_f32, int out_f32, int dtype, void* stream);
/* nn.MaxPool2d(2) (guided_fusion_model.py:78) */
int pf_maxpool2(const void* x, int x_ld, int B, int H, int W, int C, void* y, int y_ld, int dtype, void* stream);
/* channel-slice copy y[..., 0..C) = x[..., 0..C) with optional dtype change */
int pf_copy_channels(const void* x, int x_ld, void* y, int y_ld, long npix, int C, int in_f32, int out_f32,
                     int dtype, void* stream);
/* fusion-net input cat[coarse_depth_roi, fine_depth, rgb crop] (patchfusion.py:269) -> [B,h,w,8] (3 zero pad) */
int pf_pack_fusion_input(const float* cdepth, const float* fdepth, const float* crops, void* y, int B, int h, int w,
                         int dtype, void* stream);
/* NHWC (ld) <-> NCHW float converters for the module boundary */
int pf_nhwc_to_nchw_f32(const void* x, int x_ld, float* y, int B, int H, int W, int C, int in_f32, int dtype, void* stream);

/* ---- metric-bins head ----------------------------------------------------------------------------- */
/* AttractorLayerUnnormed (attractor.py:164-208) and the bounded AttractorLayer (:60-136) -- bin_centers_type, zoedepth_v1.py:90-104:
 * c = bilinear_up(b_prev); out = c + reduce_a dist(A_a - c), A_a = A[a * a_stride] + a_eps.
 *   attractor_exp 0: inv_attractor dx / (1 + 300 dx^2) (:44-57), 1: exp_attractor exp(-300 dx^2) dx (:29-41) -- alpha = 300, gamma = 2
 *   are the jit functions' DEFAULTS: the layers call dist() without their own alpha / gamma, the config values never arrive;
 *   kind_sum 0: mean over the attractors, 1: sum;
 *   unnormed layer: A = softplus(mlp), a_stride 1, a_eps 0; bounded layer: A = relu(mlp) (2 n_attr channels), a_stride 2, a_eps 1e-3
 *   (:105-106 overwrites the normalised pair with A[:, :, 0]).  All float. */
int pf_attractor(const float* A, int a_ld, int n_attr, int a_stride, float a_eps, int attractor_exp, int kind_sum,
                 const float* b_prev, int hp, int wp, float* out, int B, int h, int w, int n_bins, void* stream);
/* Seed bin centres of the bounded variants: x [npix][x_ld] float = relu(mlp) (bounded) or softplus(mlp), out [npix][n_bins].
 *   bounded   (SeedBinRegressor, localbins_layers.py:52-68): Bn = x + 1e-3; widths = (max - min) Bn / sum(Bn); edges = cumsum([min, widths]);
 *             centre_k = (edge_k + edge_k+1) / 2
 *   normalize (zoedepth_v1.py:178-182, 'normed' and 'hybrid2'): centre -> (centre - min) / (max - min) */
int pf_seed_bin_centers(const float* x, int x_ld, float* out, long npix, int n_bins, float min_depth, float max_depth, int bounded,
                        int normalize, void* stream);
/* AttractorLayer tail (attractor.py:132-135): out = clip(sort_k((max - min) * b + min), min, max) per pixel; n_bins <= 64 */
int pf_bounded_bin_centers(const float* b, float* out, long npix, int n_bins, float min_depth, float max_depth, void* stream);
/* ConditionalLogBinomial tail + expectation (dist_layers.py:29-33,51-69,108-121, zoedepth_v1.py:215-219):
 * pt [B,h,w,4] float = softplus(mlp) ; centers [B,hc,wc,n_bins] float ; depth [B,h,w] float */
int pf_logbinom_depth(const float* pt, int pt_ld, const float* centers, int hc, int wc, float* depth, int B, int h,
                      int w, int n_bins, float min_temp, float max_temp, void* stream);
/* The whole full-resolution tail of a metric-bins head in one launch (float32): cat[last(32), up(b_embedding)(128), rel?] -> Conv1x1(80) +
 * GELU -> Conv1x1(4) + Softplus -> log-binomial expectation over the up-sampled bin centres (zoedepth_v1.py:207-219, dist_layers.py:97-121).
 * clb: the CLB buffer [B,H,W,clb_ld] whose channels [0,32) hold `last` and (nq == 11) [rel_off, rel_off+8) the relative depth; emb: the
 * low-resolution embedding [B,he,we,128]; w0f / b0 / w2 / b2: packing.bins_tail_weights; centers [B,hc,wc,64]; depth [B,H,W].
 * Replaces pf_resize_bilinear (embedding into the CLB buffer) + two pf_conv + pf_logbinom_depth. */
int pf_bins_tail(const float* clb, int clb_ld, int rel_off, const float* emb, int he, int we, const float* w0f, const float* b0,
                 const float* w2, const float* b2, const float* centers, int hc, int wc, float* depth, int B, int H, int W, int nq,
                 float min_temp, float max_temp, void* stream);

/* ---- stitching (estimator/models/utils.py:21-36, baseline_pretrain.py:310-329,205-216) ------------ */
/* init pass: pred[y0+i,x0+j] = depth*mask ; count[...] = mask  (P tiles) */
int pf_stitch_init(float* pred, float* count, int MH, int MW, const float* depth, const float* mask, const int* yx,
                   int P, int ph, int pw, void* stream);
int pf_stitch_finish_init(float* avg, const float* pred, const float* count, long n, void* stream);
/* RunningAverageMap.update restricted to the tile's footprint, tiles applied in order.
 * depth tile is [dh][dw]; when (dh,dw) != (ph,pw) it is nearest-resized (F.interpolate default,
 * baseline_pretrain.py:203) on the fly. */
int pf_stitch_update(float* avg, float* count, int MH, int MW, const float* depth, int dh, int dw, const float* mask,
                     int y0, int x0, int ph, int pw, void* stream);
/* RunningAverageMap.resize: avg nearest, count bilinear align_corners */
int pf_resize_nearest_f32(const float* x, int H, int W, float* y, int OH, int OW, void* stream);
int pf_resize_bilinear_f32(const float* x, int H, int W, float* y, int OH, int OW, void* stream);

/* ==== input / output side of the path: HBM-bound byte/float streaming ==== */

/* estimator/datasets/general_dataset.py:22-47 read_image(): decoded uint8 HWC RGB -> `img / 255.0` (float64) ->
 * F.interpolate(bicubic, align_corners=True) to image_raw_shape -> float32 CHW planes (`to_tensor(...).float()`, :201).
 * Evaluated in double like the reference; H == OH && W == OW is the exact conversion; reverse_channels = the
 * `[:, :, ::-1]` of the 'u4k' raw-file branch (:24-25). dst is [3][OH][OW]. */
int pf_u8_bicubic_to_f32(const uint8_t* src, int H, int W, int reverse_channels, float* dst, int OH, int OW, void* stream);

/* estimator/utils/color.py:121-128: np.percentile(value[mask], q0 / q1) with mask = (value != invalid_val) when
 * use_invalid, or mask = (invalid_mask[i] == 0) when invalid_mask (n bytes, device) is not NULL -- an explicit mask REPLACES
 * the value test, as in the reference.  Exact order statistics by a three-level radix select on order-preserving integer keys, linear
 * interpolation as numpy 1.24 (the reference's pinned version): index and weight in double, difference in float32.
 * out2 = {percentile q0, percentile q1} (device, float32; NaN when no valid sample).  `workspace`: device buffer of
 * pf_percentile_workspace_bytes() bytes, contents irrelevant on entry. */
int pf_percentile_workspace_bytes(void);
int pf_percentiles_f32(const float* x, long n, float invalid_val, int use_invalid, const uint8_t* invalid_mask, double q0, double q1,
                       float* out2, void* workspace, void* stream);

/* estimator/utils/color.py:130-150 + matplotlib Colormap.__call__(bytes=True): x = (v - vmin)/(vmax - vmin) in float32
 * (v*0 when vmin == vmax), index = trunc(x*N) with matplotlib's under / over / bad rules; lut_rgba has N+3 RGBA rows
 * (N colours, under, over, bad), already `(lut*255).astype(uint8)`; vmin_vmax is a device float[2] (e.g. the output of
 * pf_percentiles_f32); invalid pixels (v == invalid_val, or invalid_mask[i] != 0 when the mask is not NULL) get
 * background_rgba (R | G<<8 | B<<16 | A<<24). out: [n][4].  gamma_corrected (color.py:86-91) is a per-byte map, so the host
 * applies it to the 1 KiB table and the background colour instead of to the image. */
int pf_colorize_f32(const float* depth, long n, const float* vmin_vmax, const uint8_t* lut_rgba, int N, float invalid_val,
                    int use_invalid, const uint8_t* invalid_mask, uint32_t background_rgba, uint8_t* out_rgba, void* stream);

/* estimator/tester/tester.py:75: (depth * 256).astype('uint16') (float32 product, truncation; clamped to [0, 65535]) */
int pf_depth_to_u16(const float* depth, long n, float scale, uint16_t* out, void* stream);

/* estimator/utils/metric.py:87-148 compute_metrics (+ compute_errors :10-52, soft_edge_error :66-71) as one masked
 * reduction over the ground-truth grid [H][W]: pred [ph][pw] is resized on the fly (bilinear, align_corners=False)
 * when the grids differ, clamped to [min_depth, max_depth] (inf -> max, nan -> min); mask = min < gt < max inside the
 * evaluation rectangle rows [crop_y0, crop_y1) x cols [crop_x0, crop_x1) (garg / eigen crops; whole image = 0,H,0,W);
 * edges (or NULL) = disp_gt_edges, non-zero = boundary pixel; additional_mask (or NULL) = [H][W] bytes, zero = excluded
 * (metric.py:128-130, prompt-depth evaluation).  Terms in float32 like numpy, sums in double:
 * out13 = n, #(thresh<1.25), #(<1.25^2), #(<1.25^3), S|gt-p|/gt, S(gt-p)^2/gt, S(gt-p)^2, S(ln gt - ln p)^2,
 *         S(ln p - ln gt), S(ln p - ln gt)^2, S|log10 gt - log10 p|, S soft-edge error, #(mask & edges)   (device) */
int pf_depth_metrics(const float* gt, int H, int W, const float* pred, int ph, int pw, const float* edges,
                     const uint8_t* additional_mask, float min_depth, float max_depth, int crop_y0, int crop_y1, int crop_x0, int crop_x1,
                     double* out13, void* stream);

/* SILogLoss forward value (estimator/models/losses.py:15-62), the loss of PatchFusion.forward(mode='train') (patchfusion.py:395):
 * pred / target float32 [n] of equal size; ws3 = 3 doubles of device scratch; *loss (device float) = 10*sqrt(var(g) + beta*mean(g)^2)
 * over min_depth < target < max_depth, g = log(pred+1e-7) - log(target+1e-7); 0 when <= 1 valid element. */
int pf_silog_loss(const float* pred, const float* target, long n, float min_depth, float max_depth, float beta, double* ws3,
                  float* loss, void* stream);

#ifdef __cplusplus
}
#endif
#endif
