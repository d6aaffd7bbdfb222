// Winograd F(m x m, 3 x 3) path of the float32 ("exact") mode for the large 3x3 / stride-1 / pad-1 convolutions of the DPT head and
// the guided-fusion U-Net (estimator/models/blocks/guided_fusion_model.py:41-48,85-100; external/depth_anything/dpt.py, blocks.py
// ResidualConvUnit) -- 77 % of the f32 image pass is spent in those layers (profiles/r2b_op_roofline_fp32.md).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A          (Lavin & Gray; m = 2: 2.25x, m = 4: 4x fewer multiplies than direct)
//
// Three steps per layer, the middle one on the matrix cores through the existing f32 implicit-GEMM kernel:
//   1. wino_input_kernel   x NHWC [B,H,W,C]           -> V [(m+2)^2][T][C]      T = B * ceil(H/m) * ceil(W/m) tiles  (HBM-bound)
//   2. pf_conv (1x1)       V[k] [T][C] x U[k] [N][C]  -> M[k] [T][N]           (m+2)^2 GEMM planes in ONE batched launch, U = G g G^T
//   3. wino_output_kernel  M [(m+2)^2][T][N]          -> y NHWC [B,H,W,N]  + bias, ReLU, residual(s)                 (HBM-bound)
// The transforms are exact small-integer / dyadic combinations evaluated in float32; the result differs from the direct
// convolution only by rounding (measured per layer: m = 2 ~2.5x, m = 4 ~15x the rounding error of a direct f32 convolution,
// DESIGN.md 4d).  Only float32 tensors; everything else stays on the direct kernels.
#include <mutex>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

#define ST(s) reinterpret_cast<hipStream_t>(s)

template <int MT> struct Wino;
template <> struct Wino<2> {
  // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]      A^T = [1 1 1 0; 0 1 -1 -1]
  __device__ static __forceinline__ void bt(float (&d)[4]) {
    const float o0 = d[0] - d[2], o1 = d[1] + d[2], o2 = d[2] - d[1], o3 = d[1] - d[3];
    d[0] = o0; d[1] = o1; d[2] = o2; d[3] = o3;
  }
  __device__ static __forceinline__ void at(const float (&m)[4], float (&o)[2]) {
    o[0] = m[0] + m[1] + m[2];
    o[1] = m[1] - m[2] - m[3];
  }
};
template <> struct Wino<4> {
  // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
  // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
  __device__ static __forceinline__ void bt(float (&d)[6]) {
    const float o0 = 4.f * d[0] - 5.f * d[2] + d[4];
    const float o1 = -4.f * (d[1] + d[2]) + d[3] + d[4];
    const float o2 = 4.f * (d[1] - d[2]) - d[3] + d[4];
    const float o3 = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    const float o4 = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    const float o5 = 4.f * d[1] - 5.f * d[3] + d[5];
    d[0] = o0; d[1] = o1; d[2] = o2; d[3] = o3; d[4] = o4; d[5] = o5;
  }
  __device__ static __forceinline__ void at(const float (&m)[6], float (&o)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
  }
};

// one thread = one tile x 4 channels: (m+2)^2 16-byte loads (zero outside the image = the conv's padding), B^T d B in registers,
// (m+2)^2 16-byte stores into the planes of V.  Consecutive lanes take consecutive channel groups: every load / store of a wave is
// a contiguous 1 KiB run of one pixel / one V row.
// SPLIT: V is written as the three bf16 planes of the split-precision GEMM (csrc/gemm_split3.hip), CHUNK-MAJOR: [3][(m+2)^2][C/32][T][32] bf16
// (every 32-channel K chunk of all tiles is one slab = what the GEMM's 1-KiB DMA pieces read as whole cache lines).  Thread order for that
// layout: 8 lanes = the 8 channel quads of one chunk of one tile (64 B of a slab row, 128 B of a pixel), 8 tiles per wave -> every store
// of a wave is one contiguous 512-byte run of a slab; then the chunks of the same 8 tiles.
// Grid-stride: a launch with fewer blocks than work items walks them (`total` items, blockDim.x at a time) -- the RESIDENT form (512 threads per
// block = one block per CU at these register counts, grid = the number of CUs the transform may take) that runs beside a capped persistent GEMM of
// the other tile batch (run_split3); a launch with one block per 256 items is the plain one-shot form.
// (Measured and NOT kept, round 5: the plane stores as 16-byte stores -- lane pairs trading halves through DPP quad_perm so that every store instruction
// writes a full KiB: three instructions per two transform points instead of six.  Image pass 193.0 vs 193.2 ms, the kernel's own total +3 %
// (profiles/r5_image_ab_pair.md): at full-chip occupancy the input transform is bound by HBM, not by store issue.  Confined to 64 CUs the same kernel runs
// 3x slower -- ~16 B/clk/CU of store issue -- which is why the transforms cannot hide on a CU subset beside the GEMM, profiles/r5_overlap_probe_resident.md.)
template <int MT, bool SPLIT = false>
__global__ __launch_bounds__(512) void wino_input_kernel(const float* __restrict__ x, int x_ld, int B, int H, int W, int C, int relu_in,
                                                         float* __restrict__ V, int TH, int TW, long total, long tile0, long T) {
  // (tile window, round 5: this launch transforms tiles [tile0, tile0 + T) of the layer's B * TH * TW tiles into a V of T rows -- run_split3 walks a
  // layer in windows small enough for the arena pair to stay in the 256 MB memory-side cache)
  constexpr int A = MT + 2;
  const int cv = C >> 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
  int c4;
  long tile;
  if constexpr (SPLIT) {
    const int nkc = C >> 5;                           // C % 32 == 0
    const long w64 = idx >> 6;                        // (tile octet, chunk) pairs, chunk fastest
    tile = (w64 / nkc) * 8 + ((idx >> 3) & 7);
    c4 = (int)(w64 % nkc) * 8 + (int)(idx & 7);
    if (tile >= T) continue;
  } else {
    if (idx >= T * cv) continue;
    c4 = (int)(idx % cv);
    tile = idx / cv;
  }
  const long gt = tile + tile0;                       // position of the tile in the layer
  const int tx = (int)(gt % TW), ty = (int)((gt / TW) % TH), b = (int)(gt / ((long)TW * TH));
  float d[A][A][4];
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const int y = ty * MT - 1 + i;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      const int xx = tx * MT - 1 + j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
        v = *reinterpret_cast<const float4*>(x + (((long)b * H + y) * W + xx) * x_ld + c4 * 4);
      if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      d[i][j][0] = v.x; d[i][j][1] = v.y; d[i][j][2] = v.z; d[i][j][3] = v.w;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int j = 0; j < A; ++j) {                     // B^T d : along the rows
      float col[A];
#pragma unroll
      for (int i = 0; i < A; ++i) col[i] = d[i][j][e];
      Wino<MT>::bt(col);
#pragma unroll
      for (int i = 0; i < A; ++i) d[i][j][e] = col[i];
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {                     // (B^T d) B : along the columns
      float row[A];
#pragma unroll
      for (int j = 0; j < A; ++j) row[j] = d[i][j][e];
      Wino<MT>::bt(row);
#pragma unroll
      for (int j = 0; j < A; ++j) d[i][j][e] = row[j];
    }
  }
  const size_t plane = (size_t)T * C;
  if constexpr (SPLIT) {
    bf16_t* o = reinterpret_cast<bf16_t*>(V) + ((size_t)(c4 >> 3) * T + tile) * 32 + (c4 & 7) * 4;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j) store_split3(o + (size_t)(i * A + j) * plane, (long)(A * A) * (long)plane, d[i][j]);
  } else {
    float* o = V + (size_t)tile * C + c4 * 4;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
      for (int j = 0; j < A; ++j)
        *reinterpret_cast<float4*>(o + (size_t)(i * A + j) * plane) = make_float4(d[i][j][0], d[i][j][1], d[i][j][2], d[i][j][3]);
  }
  }
}

// one thread = one tile x 4 output channels: (m+2)^2 16-byte loads from the planes of M, A^T m A, then the conv epilogue in the order of
// pf_conv (bias -> act -> + res -> + res2) and m x m 16-byte stores (bounds-checked: H, W need not be multiples of m)
template <int MT>
__global__ __launch_bounds__(512) void wino_output_kernel(const float* __restrict__ M, int N, const float* __restrict__ bias, int relu,
                                                          const float* __restrict__ res, int res_ld, const float* __restrict__ res2,
                                                          int res2_ld, float* __restrict__ y, int y_ld, int B, int H, int W, int TH,
                                                          int TW, long tile0, long T) {
  constexpr int A = MT + 2;
  const int nv = N >> 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < T * nv; idx += (long)gridDim.x * blockDim.x) {      // (grid-stride: see wino_input_kernel)
  const int n4 = (int)(idx % nv);
  const long tile = idx / nv;
  const long gt = tile + tile0;
  const int tx = (int)(gt % TW), ty = (int)((gt / TW) % TH), b = (int)(gt / ((long)TW * TH));
  const size_t plane = (size_t)T * N;
  const float* m = M + (size_t)tile * N + n4 * 4;
  float t[MT][A][4];                                  // A^T m : along the rows
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float col[4][A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(m + (size_t)(i * A + j) * plane);
      col[0][i] = v.x; col[1][i] = v.y; col[2][i] = v.z; col[3][i] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float o[MT];
      Wino<MT>::at(col[e], o);
#pragma unroll
      for (int p = 0; p < MT; ++p) t[p][j][e] = o[p];
    }
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + n4 * 4);
  const float be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
  for (int p = 0; p < MT; ++p) {
    const int oy = ty * MT + p;
    float r[4][MT];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float row[A];
#pragma unroll
      for (int j = 0; j < A; ++j) row[j] = t[p][j][e];
      Wino<MT>::at(row, r[e]);
    }
    if (oy >= H) continue;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      const int ox = tx * MT + q;
      if (ox >= W) continue;
      const long pix = ((long)b * H + oy) * W + ox;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = r[e][q] + be[e];
        if (relu) v[e] = fmaxf(v[e], 0.f);
      }
      if (res) {
        const float4 a = *reinterpret_cast<const float4*>(res + pix * res_ld + n4 * 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      }
      if (res2) {
        const float4 a = *reinterpret_cast<const float4*>(res2 + pix * res2_ld + n4 * 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      }
      *reinterpret_cast<float4*>(y + pix * y_ld + n4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  }
}

inline int launch_ok() { return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH; }

template <int MT>
int run(const pf_conv_params* p, const float* U, int u_rows, int u_kpad, float* V, float* M, hipStream_t st) {
  constexpr int A = MT + 2;
  const int TH = (p->H + MT - 1) / MT, TW = (p->W + MT - 1) / MT;
  const long T = (long)p->B * TH * TW;
  if (T > 0x7fffffffL) return PF_ERR_ARG;
  const long nin = T * (p->Cin / 4), nout = T * (p->Cout / 4);
  hipLaunchKernelGGL(wino_input_kernel<MT>, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(p->x), p->x_ld,
                     p->B, p->H, p->W, p->Cin, p->relu_in, V, TH, TW, nin, 0L, T);
  if (launch_ok() != PF_OK) return PF_ERR_LAUNCH;
  pf_conv_params q = {};
  q.x_ld = p->Cin; q.B = 1; q.H = 1; q.W = (int)T; q.Cin = p->Cin;
  q.w_rows = u_rows; q.Kpad = u_kpad;
  q.y_ld = p->Cout; q.OH = 1; q.OW = (int)T; q.Cout = p->Cout;
  q.KH = q.KW = 1; q.stride = 1; q.pad = 0; q.act = PF_ACT_NONE; q.shuffle = 1; q.dtype = PF_DTYPE_F32; q.korder = 0;
  q.x = V; q.w = U; q.y = M;                            // the (m+2)^2 transform points = planes of ONE batched GEMM launch
  q.batch = A * A;
  q.x_bstride = T * p->Cin; q.w_bstride = (long)u_rows * u_kpad; q.y_bstride = T * p->Cout;
  const int rc = pf_conv(&q, st);
  if (rc != PF_OK) return rc;
  hipLaunchKernelGGL(wino_output_kernel<MT>, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, M, p->Cout, p->bias,
                     p->act == PF_ACT_RELU ? 1 : 0, static_cast<const float*>(p->res), p->res_ld, static_cast<const float*>(p->res2),
                     p->res2_ld, static_cast<float*>(p->y), p->y_ld, p->B, p->H, p->W, TH, TW, 0L, T);
  return launch_ok();
}

// one event per device shared by every stream of the process (see run_split3); created once per device under a once_flag (two host threads may enter together)
hipEvent_t gemm_token() {
  static hipEvent_t ev[64] = {};
  static std::once_flag once[64];
  int dev = 0;
  hipGetDevice(&dev);
  std::call_once(once[dev & 63], [&] { hipEventCreateWithFlags(&ev[dev & 63], hipEventDisableTiming); });
  return ev[dev & 63];
}

// the three measurement switches of run_split3 (PF_W3_TGRID / PF_W3_GRID / PF_W3_TOKEN; all measured as "does not pay"), read from the environment ONCE
struct W3Switches { int tgrid = 0, cap = 0; bool token = false; };
const W3Switches& w3_switches() {
  static W3Switches sw;
  static std::once_flag once;
  std::call_once(once, [] {
    if (const char* s = getenv("PF_W3_TGRID")) sw.tgrid = atoi(s);
    if (const char* s = getenv("PF_W3_GRID")) sw.cap = atoi(s);
    if (const char* s = getenv("PF_W3_TOKEN")) sw.token = s[0] == '1';
  });
  return sw;
}

// F(4x4,3x3) with the transform-domain GEMM in split precision: V is written as three bf16 planes, U3 = the three planes of G g G^T
// (chunk-major [3][36][Cin/32][u_rows][32] bf16 = PackedConv.wino_u3, the split3 of packing.winograd_filters), ONE batched pf_gemm_split3 launch (plane = blockIdx.y), M float32.
int run_split3(const pf_conv_params* p, const void* U3, int u_rows, int u_kpad, void* V3, float* M, long window, hipStream_t st) {
  constexpr int MT = 4, A = 6;
  const int TH = (p->H + MT - 1) / MT, TW = (p->W + MT - 1) / MT;
  const long Tall = (long)p->B * TH * TW;
  if (Tall > 0x7fffffffL) return PF_ERR_ARG;
  if (window <= 0 || window > Tall) window = Tall;
  // PF_W3_TGRID = n > 0: the transforms run RESIDENT on n CUs (n blocks of 512 threads walking the items) instead of flooding the chip.
  // Overlap of the HBM-bound transforms of one tile batch with the matrix-bound GEMM of the other (two streams; measured, off): the
  // persistent GEMM leaves CUs free (PF_W3_GRID blocks instead of one per CU) and, with PF_W3_TOKEN=1, the big GEMMs of ALL streams are chained
  // through one event in host-issue order, so that two capped GEMMs never compete for the same CUs.
  const W3Switches& sw = w3_switches();
  const int tgrid = sw.tgrid, cap = sw.cap;
  hipEvent_t tok = sw.token ? gemm_token() : nullptr;
  // With more than one window, window k's output is written before window k + 1's input (and its one-pixel halo across the seam) is read: x and y must
  // not overlap (a single window reads all of x first, like the unwindowed sequence).  Refused instead of computing from half-overwritten input.
  if (window < Tall) {
    const char* xa = static_cast<const char*>(p->x);
    const char* ya = static_cast<const char*>(p->y);
    const long xb = ((long)p->B * p->H * p->W - 1) * p->x_ld * 4 + (long)p->Cin * 4, yb = ((long)p->B * p->H * p->W - 1) * p->y_ld * 4 + (long)p->Cout * 4;
    if (xa < ya + yb && ya < xa + xb) return PF_ERR_ARG;
  }
  // TILE WINDOWS (round 5): the layer's tiles go through the three steps `window` tiles at a time, all windows through the SAME V / M arena -- launches
  // of one stream are ordered.  Winograd tiles are independent, so the numbers do not depend on the window.
  for (long t0 = 0; t0 < Tall; t0 += window) {
    const long T = Tall - t0 < window ? Tall - t0 : window;
    const long nin = ((T + 7) / 8) * 8 * (p->Cin / 4), nout = T * (p->Cout / 4);       // (input transform: whole tile octets, see the kernel)
    const bool resident = tgrid > 0 && nin > (long)tgrid * 512 * 4;
    const dim3 gin = resident ? dim3((unsigned)tgrid) : dim3((unsigned)((nin + 255) / 256)), bin = resident ? dim3(512) : dim3(256);
    hipLaunchKernelGGL((wino_input_kernel<MT, true>), gin, bin, 0, st, static_cast<const float*>(p->x), p->x_ld,
                       p->B, p->H, p->W, p->Cin, p->relu_in, static_cast<float*>(V3), TH, TW, nin, t0, T);
    if (launch_ok() != PF_OK) return PF_ERR_LAUNCH;
    pf_conv_params q = {};
    q.x_ld = p->Cin; q.B = 1; q.H = 1; q.W = (int)T; q.Cin = p->Cin;
    q.w_rows = u_rows; q.Kpad = u_kpad;
    q.y_ld = p->Cout; q.OH = 1; q.OW = (int)T; q.Cout = p->Cout;
    q.KH = q.KW = 1; q.stride = 1; q.pad = 0; q.act = PF_ACT_NONE; q.shuffle = 1; q.dtype = PF_DTYPE_BF16; q.out_f32 = 1;
    q.x = V3; q.w = U3; q.y = M;
    q.korder = 6;                                         // V3 and U3 are chunk-major: [plane][point][Cin/32][rows][32]
    q.batch = A * A;                                      // transform points: x / w / y advance by one [T][Cin] / [rows][Kpad] / [T][Cout] block each
    q.x_bstride = (long)A * A * T * p->Cin;               // h / m / l plane strides
    q.w_bstride = (long)A * A * u_rows * u_kpad;
    if (tok) hipStreamWaitEvent(st, tok, 0);
    const int rc = pf_gemm_split3_ex(&q, cap, st);
    if (rc != PF_OK) return rc;
    if (tok) hipEventRecord(tok, st);
    const bool resident_out = tgrid > 0 && nout > (long)tgrid * 512 * 4;
    hipLaunchKernelGGL(wino_output_kernel<MT>, resident_out ? dim3((unsigned)tgrid) : dim3((unsigned)((nout + 255) / 256)), resident_out ? dim3(512) : dim3(256), 0, st, M,
                       p->Cout, p->bias, p->act == PF_ACT_RELU ? 1 : 0, static_cast<const float*>(p->res), p->res_ld, static_cast<const float*>(p->res2),
                       p->res2_ld, static_cast<float*>(p->y), p->y_ld, p->B, p->H, p->W, TH, TW, t0, T);
    if (launch_ok() != PF_OK) return PF_ERR_LAUNCH;
  }
  return PF_OK;
}

}  // namespace

extern "C" int pf_conv_winograd_split3(const pf_conv_params* p, const void* U3, int u_rows, int u_kpad, void* V3, void* M, void* stream) {
  if (!p || !U3 || !V3 || !M || !p->x || !p->y) return PF_ERR_ARG;
  if (p->dtype != PF_DTYPE_F32 || p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1 || p->shuffle > 1 || p->scale) return PF_ERR_ARG;
  if (p->OH != p->H || p->OW != p->W || p->Cin % 32 || p->Cout % 8 || p->Cin <= 0 || p->Cout <= 0) return PF_ERR_ARG;
  if (p->act != PF_ACT_NONE && p->act != PF_ACT_RELU) return PF_ERR_ARG;
  if (u_rows < p->Cout || u_kpad != p->Cin) return PF_ERR_ARG;       // (chunk-major planes are dense in K)
  return run_split3(p, U3, u_rows, u_kpad, V3, static_cast<float*>(M), 0, ST(stream));
}

// the same layer `window` Winograd tiles at a time (0 / >= all tiles: one window): V3 / M are arenas for ONE window -- 3 x 36 x ceil8(window) x Cin bf16
// and 36 x window x Cout float32.  Identical results for every window (tiles are independent).
extern "C" int pf_conv_winograd_split3_windowed(const pf_conv_params* p, const void* U3, int u_rows, int u_kpad, void* V3, void* M, long window, void* stream) {
  if (!p || !U3 || !V3 || !M || !p->x || !p->y || window < 0 || (window > 0 && window % 8)) return PF_ERR_ARG;
  if (p->dtype != PF_DTYPE_F32 || p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1 || p->shuffle > 1 || p->scale) return PF_ERR_ARG;
  if (p->OH != p->H || p->OW != p->W || p->Cin % 32 || p->Cout % 8 || p->Cin <= 0 || p->Cout <= 0) return PF_ERR_ARG;
  if (p->act != PF_ACT_NONE && p->act != PF_ACT_RELU) return PF_ERR_ARG;
  if (u_rows < p->Cout || u_kpad != p->Cin) return PF_ERR_ARG;
  return run_split3(p, U3, u_rows, u_kpad, V3, static_cast<float*>(M), window, ST(stream));
}

extern "C" int pf_conv_winograd(const pf_conv_params* p, int m, const void* U, int u_rows, int u_kpad, void* V, void* M, void* stream) {
  if (!p || !U || !V || !M || !p->x || !p->y) return PF_ERR_ARG;
  if (p->dtype != PF_DTYPE_F32 || p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1 || p->shuffle > 1 || p->scale) return PF_ERR_ARG;
  if (p->OH != p->H || p->OW != p->W || p->Cin % 32 || p->Cout % 8 || p->Cin <= 0 || p->Cout <= 0) return PF_ERR_ARG;
  if (p->act != PF_ACT_NONE && p->act != PF_ACT_RELU) return PF_ERR_ARG;
  if (u_rows < p->Cout || u_kpad < p->Cin || u_kpad % 32) return PF_ERR_ARG;
  if (m == 2) return run<2>(p, static_cast<const float*>(U), u_rows, u_kpad, static_cast<float*>(V), static_cast<float*>(M), ST(stream));
  if (m == 4) return run<4>(p, static_cast<const float*>(U), u_rows, u_kpad, static_cast<float*>(V), static_cast<float*>(M), ST(stream));
  return PF_ERR_ARG;
}
