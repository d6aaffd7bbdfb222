// HBM-bound image kernels of the PatchFusion hot path: bilinear resize (align_corners=True),
// per-tile crop+resize, roi_align, max-pool, channel-slice copy, fusion-net input packing, layout
// conversion, the metric-bins head tail (attractor, log-binomial expectation) and the tile stitcher.
// All loads/stores are 16-byte vectors over the contiguous channel axis of NHWC tensors.
#include <atomic>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
inline int ok() { return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH; }

// PyTorch upsample_bilinear2d, align_corners=True: scale = (in-1)/(out-1); src = scale*dst
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp ac_coord(int dst, float scale, int in) {
#pragma clang fp contract(off)      // l1 = fl(scale * dst) - i0 like the reference's CPU evaluation; a fused multiply-subtract would differ by an ulp
  const float src = scale * dst;
  Lerp r;
  r.i0 = (int)src;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = src - r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}
// the bilinear blend, one evaluation order for every resize kernel (no contraction: the kernels agree bit for bit)
__device__ __forceinline__ float bilerp(const Lerp& ly, const Lerp& lx, float v00, float v01, float v10, float v11) {
#pragma clang fp contract(off)
  return ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
}
inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
__device__ __forceinline__ void ld8x(const void* p, long off, int f32, float (&v)[8]) {
  if (f32) load8(reinterpret_cast<const float*>(p) + off, v);
  else load8(reinterpret_cast<const T*>(p) + off, v);
}
template <typename T>
__device__ __forceinline__ void st8x(void* p, long off, int f32, const float (&v)[8]) {
  if (f32) store8(reinterpret_cast<float*>(p) + off, v);
  else store8(reinterpret_cast<T*>(p) + off, v);
}

// One (batch, output row) per blockIdx.y: the vertical taps are block-uniform (scalar), the loop index decomposes into
// (column, channel vector) with one 32-bit division by the channel-vector count -- the old flat grid-stride loop spent its
// time in three 64-bit divisions per 16-byte store (27-37 % of the HBM roofline, profiles/r2a_op_roofline_*.md).
template <typename T>
__global__ void resize_bilinear_kernel(const void* __restrict__ x, int x_ld, int B, int H, int W, int C, void* __restrict__ y,
                                       int y_ld, int OH, int OW, const void* __restrict__ add, int add_ld, int in_f32,
                                       int out_f32, float sh, float sw) {
  const int cv = C >> 3;
  const int n = OW * cv;
  for (int row = blockIdx.y; row < B * OH; row += gridDim.y) {      // row = b * OH + oy
  const int b = row / OH, oy = row - b * OH;
  const Lerp ly = ac_coord(oy, sh, H);
  const long r0 = ((long)b * H + ly.i0) * W, r1 = ((long)b * H + ly.i1) * W;
  const long orow = (long)row * OW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int ox = i / cv, v = i - ox * cv;
    const Lerp lx = ac_coord(ox, sw, W);
    float v00[8], v01[8], v10[8], v11[8], o[8];
    ld8x<T>(x, (r0 + lx.i0) * x_ld + v * 8, in_f32, v00);
    ld8x<T>(x, (r0 + lx.i1) * x_ld + v * 8, in_f32, v01);
    ld8x<T>(x, (r1 + lx.i0) * x_ld + v * 8, in_f32, v10);
    ld8x<T>(x, (r1 + lx.i1) * x_ld + v * 8, in_f32, v11);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bilerp(ly, lx, v00[e], v01[e], v10[e], v11[e]);
    const long pix = orow + ox;
    if (add) {
      float a[8];
      ld8x<T>(add, pix * add_ld + v * 8, out_f32, a);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = a[e] + o[e];
    }
    st8x<T>(y, pix * y_ld + v * 8, out_f32, o);
  }
  }
}

// Upv1 concat buffer in ONE launch (guided_fusion_model.py:96-99: cat[feat_enc, up(cat[temp, guide])] ): up to three
// bilinear (align_corners=True) resizes of different sources / source sizes into consecutive channel ranges of the same
// NHWC output rows.
struct ResizeSrc {
  const void* x;
  int ld, H, W, C;
  float sh, sw;
};
template <typename T>
__global__ void resize_concat_kernel(ResizeSrc s0, ResizeSrc s1, ResizeSrc s2, int nsrc, int B, void* __restrict__ y, int y_ld,
                                     int OH, int OW) {
  // one (batch, output row) per blockIdx.y; the sources are walked one after the other (block-uniform: the source
  // descriptor, its vertical taps and its channel offset stay scalar -- a per-thread source select cost more VALU than the
  // whole-row stores saved, profiles/r2b_op_roofline_fp32.md)
  for (int row = blockIdx.y; row < B * OH; row += gridDim.y) {      // row = b * OH + oy
    const int b = row / OH, oy = row - b * OH;
    const long orow = (long)row * OW;
    int coff = 0;
    for (int si = 0; si < nsrc; ++si) {
      const ResizeSrc& s = si == 0 ? s0 : (si == 1 ? s1 : s2);
      const int cv = s.C >> 3;
      const int n = OW * cv;
      const Lerp ly = ac_coord(oy, s.sh, s.H);
      const long r0 = ((long)b * s.H + ly.i0) * s.W, r1 = ((long)b * s.H + ly.i1) * s.W;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ox = i / cv, v = i - ox * cv;
        const Lerp lx = ac_coord(ox, s.sw, s.W);
        float v00[8], v01[8], v10[8], v11[8], o[8];
        ld8x<T>(s.x, (r0 + lx.i0) * s.ld + v * 8, 0, v00);
        ld8x<T>(s.x, (r0 + lx.i1) * s.ld + v * 8, 0, v01);
        ld8x<T>(s.x, (r1 + lx.i0) * s.ld + v * 8, 0, v10);
        ld8x<T>(s.x, (r1 + lx.i1) * s.ld + v * 8, 0, v11);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bilerp(ly, lx, v00[e], v01[e], v10[e], v11[e]);
        st8x<T>(y, (orow + ox) * y_ld + coff + v * 8, 0, o);
      }
      coff += s.C;
    }
  }
}

// ---- source-aligned bilinear resize (version 2) -------------------------------------------------------------------------------------
// The kernels above walk the OUTPUT pixels and fetch four taps each: an upsampling (every use in the image pass: x1.75 / x2) re-fetches
// every source pixel ~(scale)^2 x 4 times and sits at 31-40 % of the HBM roofline, bound by load issue, not bytes (profiles/r3_op_roofline).
// Here a block owns one SOURCE row interval (b, r) and a thread one source column interval c for one 16-byte channel vector: the four taps
// x[r..r+1][c..c+1] are loaded ONCE and every output pixel whose (i0y, i0x) == (r, c) is produced from them -- ~1.3 loads per output
// instead of 4, no per-output index division.  The outputs of an interval are found with the SAME float expression the output-walking
// kernels use ((int)(scale * o)), so every output pixel is written exactly once with bit-identical taps, weights and evaluation order
// (any scale, down-sampling included: intervals without outputs simply write nothing).
__device__ __forceinline__ int first_dst(int c, float scale, int out) {      // smallest o in [0, out] with (int)(scale * o) >= c
  if (c <= 0) return 0;
  if (scale <= 0.f) return out;
  int o = (int)((float)c / scale);
  o = o < 0 ? 0 : (o > out ? out : o);
  while (o > 0 && (int)(scale * (o - 1)) >= c) --o;
  while (o < out && (int)(scale * o) < c) ++o;
  return o;
}

template <typename T>
__device__ __forceinline__ void ldv(const T* p, float (&v)[16 / sizeof(T)]) {
  if constexpr (sizeof(T) == 4) load4(p, v);
  else load8(p, v);
}
template <typename T>
__device__ __forceinline__ void stv(T* p, const float (&v)[16 / sizeof(T)]) {
  if constexpr (sizeof(T) == 4) store4(p, v[0], v[1], v[2], v[3]);
  else store8(p, v);
}

// Block: RPB consecutive source rows (b, r .. r + RPB - 1) of every source.  Per source the block first tabulates, in LDS, the output-column
// range of every source column interval and the horizontal weights of every output column (they depend on neither b, r nor the channel).
// A thread then owns a (column interval, channel vector) and walks DOWN its rows: the bottom taps of one row are the top taps of the next, so
// a row costs two 16-byte loads (2.25 per row with the first), and per output pixel one LDS read, the blend and one 16-byte store.
constexpr int RESIZE_RPB = 8;
template <typename T>
__global__ void resize_src_kernel(ResizeSrc s0, ResizeSrc s1, ResizeSrc s2, int nsrc, int B, int Hmax, void* __restrict__ yv, int y_ld, int OH,
                                  int OW, const void* __restrict__ addv, int add_ld, int Wmax, int rpb) {
  constexpr int N = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char resize_lds[];
  int2* rowrange = reinterpret_cast<int2*>(resize_lds);                                  // [RESIZE_RPB] output rows [x, y) of source row r
  int2* colrange = reinterpret_cast<int2*>(resize_lds + RESIZE_RPB * 8);                 // [Wmax]  output columns [x, y) of column interval c
  float2* wx = reinterpret_cast<float2*>(resize_lds + RESIZE_RPB * 8 + (size_t)Wmax * 8);  // [OW]    (l0, l1) of output column ox
  T* __restrict__ y = reinterpret_cast<T*>(yv);
  const T* __restrict__ add = reinterpret_cast<const T*>(addv);
  const int tid = threadIdx.x, nthr = blockDim.x * gridDim.x, gtid = blockIdx.x * blockDim.x + tid;
  const int gpi = (Hmax + rpb - 1) / rpb;                                                  // row groups per image
  int coff = 0;
  for (int si = 0; si < nsrc; ++si) {
    const ResizeSrc& s = si == 0 ? s0 : (si == 1 ? s1 : s2);
    __syncthreads();
    for (int c = tid; c < s.W; c += blockDim.x) colrange[c] = make_int2(first_dst(c, s.sw, OW), first_dst(c + 1, s.sw, OW));
    for (int ox = tid; ox < OW; ox += blockDim.x) {
      const Lerp lx = ac_coord(ox, s.sw, s.W);
      wx[ox] = make_float2(lx.l0, lx.l1);
    }
    const T* __restrict__ x = reinterpret_cast<const T*>(s.x);
    const int cv = s.C / N, n = s.W * cv;
    const float inv_cv = 1.0f / (float)cv;
    for (int g = blockIdx.y; g < B * gpi; g += gridDim.y) {              // block-uniform
      const int b = g / gpi, rbeg = (g - b * gpi) * rpb;
      const int rend = rbeg + rpb < s.H ? rbeg + rpb : s.H;
      __syncthreads();                                                  // (also publishes the column tables on the first pass)
      if (tid < rpb) rowrange[tid] = make_int2(first_dst(rbeg + tid, s.sh, OH), first_dst(rbeg + tid + 1, s.sh, OH));
      __syncthreads();
      if (rbeg >= rend) continue;
      for (int i = gtid; i < n; i += nthr) {
        int c = (int)(((float)i + 0.5f) * inv_cv);                       // i / cv for i < 2^22 (the host checks)
        int v = i - c * cv;
        if (v < 0) { --c; v += cv; } else if (v >= cv) { ++c; v -= cv; }
        const int2 rg = colrange[c];
        if (rg.x >= rg.y) continue;
        const long o0 = (long)c * s.ld + v * N, o1 = (long)(c + (c < s.W - 1 ? 1 : 0)) * s.ld + v * N;
        const T* __restrict__ rowp = x + ((long)b * s.H + rbeg) * s.W * s.ld;
        float v00[N], v01[N], v10[N], v11[N];
        ldv<T>(rowp + o0, v00);
        ldv<T>(rowp + o1, v01);
        for (int r = rbeg; r < rend; ++r) {
          if (r < s.H - 1) {
            rowp += (long)s.W * s.ld;
            ldv<T>(rowp + o0, v10);
            ldv<T>(rowp + o1, v11);
          } else {
#pragma unroll
            for (int e = 0; e < N; ++e) { v10[e] = v00[e]; v11[e] = v01[e]; }
          }
          const int2 rr = rowrange[r - rbeg];
          for (int oy = rr.x; oy < rr.y; ++oy) {
            const Lerp ly = ac_coord(oy, s.sh, s.H);
            const long pix0 = ((long)b * OH + oy) * OW;
            T* __restrict__ yp = y + (pix0 + rg.x) * y_ld + coff + v * N;
            for (int ox = rg.x; ox < rg.y; ++ox, yp += y_ld) {
              const float2 w = wx[ox];
              Lerp lx;
              lx.l0 = w.x;
              lx.l1 = w.y;
              float o[N];
#pragma unroll
              for (int e = 0; e < N; ++e) o[e] = bilerp(ly, lx, v00[e], v01[e], v10[e], v11[e]);
              if (add) {
                float a[N];
                ldv<T>(add + (pix0 + ox) * add_ld + v * N, a);
#pragma unroll
                for (int e = 0; e < N; ++e) o[e] = a[e] + o[e];
              }
              stv<T>(yp, o);
            }
          }
#pragma unroll
          for (int e = 0; e < N; ++e) { v00[e] = v10[e]; v01[e] = v11[e]; }
        }
      }
    }
    coff += s.C;
  }
}

__global__ void crop_resize_planar_kernel(const float* __restrict__ img, int C, int H, int W, const int* __restrict__ boxes, int P,
                                          float* __restrict__ out, int oh, int ow) {
  const long total = (long)P * C * oh * ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int c = (int)((i / ((long)ow * oh)) % C);
    const int p = (int)(i / ((long)ow * oh * C));
    const int x0 = boxes[p * 4 + 0], y0 = boxes[p * 4 + 1], bw = boxes[p * 4 + 2] - x0, bh = boxes[p * 4 + 3] - y0;
    const float sh = oh > 1 ? (float)(bh - 1) / (float)(oh - 1) : 0.f;
    const float sw = ow > 1 ? (float)(bw - 1) / (float)(ow - 1) : 0.f;
    const Lerp ly = ac_coord(oy, sh, bh), lx = ac_coord(ox, sw, bw);
    const float* src = img + (long)c * H * W;
    const float v00 = src[(long)(y0 + ly.i0) * W + x0 + lx.i0], v01 = src[(long)(y0 + ly.i0) * W + x0 + lx.i1];
    const float v10 = src[(long)(y0 + ly.i1) * W + x0 + lx.i0], v11 = src[(long)(y0 + ly.i1) * W + x0 + lx.i1];
    out[i] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
  }
}

// torchvision roi_align, aligned=True, sampling_ratio=-1 (adaptive).  Coordinates follow the
// torchvision kernel's float32 operation order; __f*_rn intrinsics forbid FMA contraction.
// Row-wise form (round 4): block (k, ph) = one output row of one ROI; the ROI geometry and the row's sample rows are computed once per block,
// threads walk the row's (column, 8-channel vector) items with 32-bit arithmetic.  (The flat form spent its time in four 64-bit divisions per
// 32-byte store: 2.9 TB/s = 36 % of HBM on a kernel that reads a 1/16 region and only has to stream its output.)
template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(const void* __restrict__ feat, int f_ld, int Bf, int H, int W, int C,
                                                        const float* __restrict__ rois, int K, void* __restrict__ y, int y_ld, int oh, int ow,
                                                        float scale, int in_f32, int out_f32) {
  const int cv = (C + 7) >> 3;
  const int k = blockIdx.x / oh, ph = blockIdx.x - k * oh;
  const float* r = rois + k * 5;
  const int b = (int)r[0];
  const float start_w = __fsub_rn(__fmul_rn(r[1], scale), 0.5f), start_h = __fsub_rn(__fmul_rn(r[2], scale), 0.5f);
  const float end_w = __fsub_rn(__fmul_rn(r[3], scale), 0.5f), end_h = __fsub_rn(__fmul_rn(r[4], scale), 0.5f);
  const float roi_w = __fsub_rn(end_w, start_w), roi_h = __fsub_rn(end_h, start_h);
  const float bin_h = __fdiv_rn(roi_h, (float)oh), bin_w = __fdiv_rn(roi_w, (float)ow);
  const int gh = (int)ceilf(__fdiv_rn(roi_h, (float)oh)), gw = (int)ceilf(__fdiv_rn(roi_w, (float)ow));
  const float count = (float)max(gh * gw, 1);
  const long base = (long)b * H * W;
  const long row_pix = ((long)k * oh + ph) * ow;
  const float y_row = __fadd_rn(start_h, __fmul_rn((float)ph, bin_h));
  // a thread owns RUN consecutive output columns of one 8-channel vector and keeps the four tap vectors of the last sample: with the up-sampling
  // ROIs of this path (a 1/16 region onto the full map: four output columns per source column, one sample per bin) three of four samples reuse
  // them -- the flat form re-fetched 4 KB of taps through the vector cache for every KB it stored, and THAT bounded it, not HBM
  constexpr int RUN = 4;
  const int groups = (ow + RUN - 1) / RUN;
  const int items = groups * cv;
  for (int i = threadIdx.x; i < items; i += blockDim.x) {
    const int pg = i / cv, v = i - pg * cv;
    int c_yl = -1, c_xl = -1, c_yh = -1, c_xh = -1;
    float v1[8], v2[8], v3[8], v4[8];
#pragma unroll
    for (int q = 0; q < RUN; ++q) {
      const int pw = pg * RUN + q;
      if (pw >= ow) break;
      const float x_col = __fadd_rn(start_w, __fmul_rn((float)pw, bin_w));
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float yy = __fadd_rn(y_row, __fdiv_rn(__fmul_rn((float)iy + 0.5f, bin_h), (float)gh));
        for (int ix = 0; ix < gw; ++ix) {
          float xx = __fadd_rn(x_col, __fdiv_rn(__fmul_rn((float)ix + 0.5f, bin_w), (float)gw));
          float y2 = yy;
          if (y2 < -1.0f || y2 > (float)H || xx < -1.0f || xx > (float)W) continue;
          if (y2 <= 0.f) y2 = 0.f;
          if (xx <= 0.f) xx = 0.f;
          int yl = (int)y2, xl = (int)xx, yh, xh;
          if (yl >= H - 1) { yh = yl = H - 1; y2 = (float)yl; } else yh = yl + 1;
          if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
          const float ly = y2 - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          if (yl != c_yl || xl != c_xl || yh != c_yh || xh != c_xh) {
            ld8x<T>(feat, (base + (long)yl * W + xl) * f_ld + v * 8, in_f32, v1);
            ld8x<T>(feat, (base + (long)yl * W + xh) * f_ld + v * 8, in_f32, v2);
            ld8x<T>(feat, (base + (long)yh * W + xl) * f_ld + v * 8, in_f32, v3);
            ld8x<T>(feat, (base + (long)yh * W + xh) * f_ld + v * 8, in_f32, v4);
            c_yl = yl; c_xl = xl; c_yh = yh; c_xh = xh;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
        }
      }
      if (count != 1.0f) {                              // (x / 1 == x: the up-sampling ROIs of this path take one sample per bin)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] /= count;
      }
      st8x<T>(y, (row_pix + pw) * y_ld + v * 8, out_f32, acc);
    }
  }
}

// single-channel float map (the coarse depth): same sampling, scalar loads
__global__ void roi_align_scalar_kernel(const float* __restrict__ feat, int Bf, int H, int W, const float* __restrict__ rois, int K,
                                        float* __restrict__ y, int oh, int ow, float scale) {
  const long total = (long)K * oh * ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(i % ow);
    const int ph = (int)((i / ow) % oh);
    const int k = (int)(i / ((long)ow * oh));
    const float* r = rois + k * 5;
    const int b = (int)r[0];
    const float start_w = __fsub_rn(__fmul_rn(r[1], scale), 0.5f), start_h = __fsub_rn(__fmul_rn(r[2], scale), 0.5f);
    const float end_w = __fsub_rn(__fmul_rn(r[3], scale), 0.5f), end_h = __fsub_rn(__fmul_rn(r[4], scale), 0.5f);
    const float roi_w = __fsub_rn(end_w, start_w), roi_h = __fsub_rn(end_h, start_h);
    const float bin_h = __fdiv_rn(roi_h, (float)oh), bin_w = __fdiv_rn(roi_w, (float)ow);
    const int gh = (int)ceilf(__fdiv_rn(roi_h, (float)oh)), gw = (int)ceilf(__fdiv_rn(roi_w, (float)ow));
    const float count = (float)max(gh * gw, 1);
    const float* src = feat + (long)b * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      const float yy0 = __fadd_rn(__fadd_rn(start_h, __fmul_rn((float)ph, bin_h)), __fdiv_rn(__fmul_rn((float)iy + 0.5f, bin_h), (float)gh));
      for (int ix = 0; ix < gw; ++ix) {
        float xx = __fadd_rn(__fadd_rn(start_w, __fmul_rn((float)pw, bin_w)), __fdiv_rn(__fmul_rn((float)ix + 0.5f, bin_w), (float)gw));
        float y2 = yy0;
        if (y2 < -1.0f || y2 > (float)H || xx < -1.0f || xx > (float)W) continue;
        if (y2 <= 0.f) y2 = 0.f;
        if (xx <= 0.f) xx = 0.f;
        int yl = (int)y2, xl = (int)xx, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y2 = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
        const float ly = y2 - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
        acc += hy * hx * src[(long)yl * W + xl] + hy * lx * src[(long)yl * W + xh] + ly * hx * src[(long)yh * W + xl] + ly * lx * src[(long)yh * W + xh];
      }
    }
    y[i] = acc / count;
  }
}

template <typename T>
__global__ void maxpool2_kernel(const T* __restrict__ x, int x_ld, int B, int H, int W, int C, T* __restrict__ y, int y_ld) {
  const int OH = H / 2, OW = W / 2, cv = C >> 3;
  const long total = (long)B * OH * OW * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    long pix = i / cv;
    const int ox = (int)(pix % OW);
    const int oy = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((long)OW * OH));
    const long base = ((long)b * H + 2 * oy) * W + 2 * ox;
    float a[8], t[8];
    load8(x + base * x_ld + v * 8, a);
    load8(x + (base + 1) * x_ld + v * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], t[e]);
    load8(x + (base + W) * x_ld + v * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], t[e]);
    load8(x + (base + W + 1) * x_ld + v * 8, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], t[e]);
    store8(y + pix * y_ld + v * 8, a);
  }
}

template <typename T>
__global__ void copy_channels_kernel(const void* __restrict__ x, int x_ld, void* __restrict__ y, int y_ld, long npix, int C,
                                     int in_f32, int out_f32) {
  const int cv = C >> 3;
  const long total = npix * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    const long pix = i / cv;
    float a[8];
    ld8x<T>(x, pix * x_ld + v * 8, in_f32, a);
    st8x<T>(y, pix * y_ld + v * 8, out_f32, a);
  }
}

template <typename T>
__global__ void pack_fusion_input_kernel(const float* __restrict__ cd, const float* __restrict__ fd, const float* __restrict__ crops,
                                         T* __restrict__ y, int B, int h, int w) {
  const long hw = (long)h * w, total = (long)B * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / hw, p = i % hw;
    float o[8] = {cd[i], fd[i], crops[(b * 3 + 0) * hw + p], crops[(b * 3 + 1) * hw + p], crops[(b * 3 + 2) * hw + p], 0.f, 0.f, 0.f};
    store8(y + i * 8, o);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ x, int x_ld, float* __restrict__ y, int B, int H, int W, int C, int in_f32) {
  const long hw = (long)H * W, total = (long)B * C * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % hw;
    const int c = (int)((i / hw) % C);
    const long b = i / (hw * C);
    const long src = (b * hw + p) * x_ld + c;
    y[i] = in_f32 ? reinterpret_cast<const float*>(x)[src] : Elem<T>::ld(reinterpret_cast<const T*>(x) + src);
  }
}

// ---------------------------------------------------------------------------------------------
// metric-bins head
// ---------------------------------------------------------------------------------------------
// AttractorLayerUnnormed (attractor.py:164-208) and AttractorLayer (:60-136): c = bilinear_up(b_prev); out = c + reduce_a dist(A_a - c).
// EXP = exp_attractor (:29-41) instead of inv_attractor (:44-57); both with the jit defaults alpha = 300, gamma = 2 because the layers
// call dist() without their own alpha / gamma.  a_stride / a_eps: the bounded layer's attractor points are the EVEN MLP outputs
// (ReLU + 1e-3; :105-106 overwrites the normalised pair with A[:, :, 0]).  scale = 1/n_attr (kind 'mean') or 1 (kind 'sum').
template <bool EXP>
__global__ void attractor_kernel(const float* __restrict__ A, int a_ld, int n_attr, int a_stride, float a_eps, float scale,
                                 const float* __restrict__ bp, int hp, int wp, float* __restrict__ out, int B, int h, int w, int n_bins,
                                 float sh, float sw) {
  const int bv = n_bins >> 2;
  const long total = (long)B * h * w * bv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % bv);
    long pix = i / bv;
    const int ox = (int)(pix % w);
    const int oy = (int)((pix / w) % h);
    const int b = (int)(pix / ((long)w * h));
    const Lerp ly = ac_coord(oy, sh, hp), lx = ac_coord(ox, sw, wp);
    const long base = (long)b * hp * wp;
    float v00[4], v01[4], v10[4], v11[4], c[4], d[4];
    load4(bp + (base + (long)ly.i0 * wp + lx.i0) * n_bins + v * 4, v00);
    load4(bp + (base + (long)ly.i0 * wp + lx.i1) * n_bins + v * 4, v01);
    load4(bp + (base + (long)ly.i1 * wp + lx.i0) * n_bins + v * 4, v10);
    load4(bp + (base + (long)ly.i1 * wp + lx.i1) * n_bins + v * 4, v11);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c[e] = ly.l0 * (lx.l0 * v00[e] + lx.l1 * v01[e]) + ly.l1 * (lx.l0 * v10[e] + lx.l1 * v11[e]);
      d[e] = 0.f;
    }
    const float* a = A + pix * a_ld;
    for (int k = 0; k < n_attr; ++k) {
      const float ak = a[k * a_stride] + a_eps;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dx = ak - c[e];
        d[e] += EXP ? expf(-300.0f * (dx * dx)) * dx : dx / (1.0f + 300.0f * (dx * dx));
      }
    }
    store4(out + pix * n_bins + v * 4, c[0] + d[0] * scale, c[1] + d[1] * scale, c[2] + d[2] * scale, c[3] + d[3] * scale);
  }
}

// SeedBinRegressor (localbins_layers.py:52-68, bounded seed centres) and the (x - min) / (max - min) of zoedepth_v1.py:178-182:
//   bounded:   Bn = relu_out + 1e-3; width_k = (max - min) * Bn_k / sum(Bn); edges = cumsum([min, width...]); centre_k = (e_k + e_k+1) / 2
//   normalize: centre -> (centre - min) / (max - min)
// One thread per pixel, channels walked in order (torch.cumsum / sum over dim 1 accumulate in this order too).
__global__ void seed_bin_centers_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, long npix, int n_bins, float lo,
                                        float hi, int bounded, int normalize) {
  const float range = hi - lo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float* r = x + i * ld;
    float* o = out + i * n_bins;
    if (bounded) {
      float sum = 0.f;
      for (int k = 0; k < n_bins; ++k) sum += r[k] + 1e-3f;
      float e0 = lo;
      for (int k = 0; k < n_bins; ++k) {
        const float e1 = e0 + range * ((r[k] + 1e-3f) / sum);
        const float c = 0.5f * (e0 + e1);
        o[k] = normalize ? (c - lo) / range : c;
        e0 = e1;
      }
    } else {
      for (int k = 0; k < n_bins; ++k) o[k] = normalize ? (r[k] - lo) / range : r[k];
    }
  }
}

// AttractorLayer tail (attractor.py:132-135): B_centers = clip(sort((max - min) * b + min)).  One wave per pixel, lane k = bin k
// (n_bins <= 64, +inf padding), bitonic sort across the lanes.
__global__ __launch_bounds__(256) void bounded_centers_kernel(const float* __restrict__ b, float* __restrict__ out, long npix, int n_bins,
                                                              float lo, float hi) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long i = wave; i < npix; i += nwaves) {
    float v = lane < n_bins ? (hi - lo) * b[i * n_bins + lane] + lo : INFINITY;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const float o = __shfl_xor(v, j, 64);
        const bool up = (lane & k) == 0, low = (lane & j) == 0;
        v = (up == low) ? fminf(v, o) : fmaxf(v, o);
      }
    }
    if (lane < n_bins) out[i * n_bins + lane] = fminf(fmaxf(v, lo), hi);
  }
}

__global__ void logbinom_depth_kernel(const float* __restrict__ pt, int pt_ld, const float* __restrict__ cen, int hc, int wc,
                                      float* __restrict__ depth, int B, int h, int w, int n_bins, float min_temp, float max_temp,
                                      float sh, float sw) {
  const long total = (long)B * h * w;
  const float eps = 1e-7f;
  const float n_ = (float)(n_bins - 1) + eps;
  const float nlogn = n_ * logf(n_);
  // log C(n, k) (Stirling form, dist_layers.py:29-45) depends on k only: tabulated once per block instead of three logf per bin, pass and
  // pixel (the kernel was bound by those: 7 % of the HBM roofline); same expression -> same values
  __shared__ float logc_tab[256];
  for (int k = threadIdx.x; k < n_bins; k += blockDim.x) {
    const float k_ = (float)k + eps;
    logc_tab[k] = nlogn - k_ * logf(k_) - (n_ - k_) * logf(n_ - k_ + eps);
  }
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % w);
    const int oy = (int)((i / w) % h);
    const int b = (int)(i / ((long)w * h));
    const float* q = pt + i * pt_ld;
    const float p0 = q[0] + 1e-4f, p1 = q[1] + 1e-4f, t0 = q[2] + 1e-4f, t1 = q[3] + 1e-4f;
    float p = p0 / (p0 + p1);
    float t = t0 / (t0 + t1);
    t = (max_temp - min_temp) * t + min_temp;
    const float omp = fminf(fmaxf(1.0f - p, 1e-4f), 1.0f);
    p = fminf(fmaxf(p, 1e-4f), 1.0f);
    const float lp = logf(p), lq = logf(omp);
    // pass 1: max_k y_k / t
    float mx = -INFINITY;
    for (int k = 0; k < n_bins; ++k) {
      const float yk = (logc_tab[k] + (float)k * lp + (float)(n_bins - 1 - k) * lq) / t;
      mx = fmaxf(mx, yk);
    }
    const Lerp ly = ac_coord(oy, sh, hc), lx = ac_coord(ox, sw, wc);
    const long base = (long)b * hc * wc;
    const float* c00 = cen + (base + (long)ly.i0 * wc + lx.i0) * n_bins;
    const float* c01 = cen + (base + (long)ly.i0 * wc + lx.i1) * n_bins;
    const float* c10 = cen + (base + (long)ly.i1 * wc + lx.i0) * n_bins;
    const float* c11 = cen + (base + (long)ly.i1 * wc + lx.i1) * n_bins;
    float num = 0.f, den = 0.f;
    for (int k4 = 0; k4 < n_bins; k4 += 4) {
      float a00[4], a01[4], a10[4], a11[4];
      load4(c00 + k4, a00); load4(c01 + k4, a01); load4(c10 + k4, a10); load4(c11 + k4, a11);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k4 + e;
        const float yk = (logc_tab[k] + (float)k * lp + (float)(n_bins - 1 - k) * lq) / t;
        const float pe = expf(yk - mx);
        const float c = ly.l0 * (lx.l0 * a00[e] + lx.l1 * a01[e]) + ly.l1 * (lx.l0 * a10[e] + lx.l1 * a11[e]);
        num += pe * c;
        den += pe;
      }
    }
    depth[i] = num / den;
  }
}

// ---------------------------------------------------------------------------------------------
// stitching
// ---------------------------------------------------------------------------------------------
__global__ void stitch_init_kernel(float* __restrict__ pred, float* __restrict__ count, int MH, int MW, const float* __restrict__ depth,
                                   const float* __restrict__ mask, const int* __restrict__ yx, int P, int ph, int pw) {
  const long total = (long)P * ph * pw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % pw);
    const int y = (int)((i / pw) % ph);
    const int p = (int)(i / ((long)pw * ph));
    const int Y = yx[2 * p] + y, X = yx[2 * p + 1] + x;
    if (Y < MH && X < MW) {
      const float m = mask[(long)y * pw + x];
      pred[(long)Y * MW + X] = depth[i] * m;
      count[(long)Y * MW + X] = m;
    }
  }
}
__global__ void div_kernel(float* __restrict__ avg, const float* __restrict__ pred, const float* __restrict__ count, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) avg[i] = pred[i] / count[i];
}
__global__ void stitch_update_kernel(float* __restrict__ avg, float* __restrict__ count, int MH, int MW, const float* __restrict__ depth,
                                     int dh, int dw, const float* __restrict__ mask, int y0, int x0, int ph, int pw, float sy, float sx) {
  const long total = (long)ph * pw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % pw), y = (int)(i / pw);
    const int Y = y0 + y, X = x0 + x;
    if (Y >= MH || X >= MW) continue;
    float d;
    if (dh == ph && dw == pw) d = depth[i];
    else {  // F.interpolate(mode='nearest'): src = min(floor(dst * in/out), in-1)
      const int syi = min((int)floorf(y * sy), dh - 1), sxi = min((int)floorf(x * sx), dw - 1);
      d = depth[(long)syi * dw + sxi];
    }
    const float m = mask[i];
    const long o = (long)Y * MW + X;
    const float c = count[o], a = avg[o];
    avg[o] = (d * m + c * a) / (c + m);
    count[o] = c + m;
  }
}
__global__ void resize_nearest_kernel(const float* __restrict__ x, int H, int W, float* __restrict__ y, int OH, int OW, float sy, float sx) {
  const long total = (long)OH * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)(i / OW);
    const int iy = min((int)floorf(oy * sy), H - 1), ix = min((int)floorf(ox * sx), W - 1);
    y[i] = x[(long)iy * W + ix];
  }
}
__global__ void resize_bilinear_f32_kernel(const float* __restrict__ x, int H, int W, float* __restrict__ y, int OH, int OW, float sh, float sw) {
  const long total = (long)OH * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)(i / OW);
    const Lerp ly = ac_coord(oy, sh, H), lx = ac_coord(ox, sw, W);
    const float v00 = x[(long)ly.i0 * W + lx.i0], v01 = x[(long)ly.i0 * W + lx.i1], v10 = x[(long)ly.i1 * W + lx.i0], v11 = x[(long)ly.i1 * W + lx.i1];
    y[i] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Metric-bins head tail at full resolution, one launch (float32): the conditional log-binomial of zoedepth_v1.py:207-219 /
// dist_layers.py:97-121 --  x = cat[last(32), rel?, up(b_embedding)(128)] -> Conv1x1(->80) + GELU -> Conv1x1(->4) + Softplus -> p, t ->
// log-binomial softmax over 64 bins -> expectation over the bilinearly up-sampled bin centres.  Replaces four launches (resize of the
// embedding into the CLB buffer, two pf_conv, pf_logbinom_depth) and their 160 / 80 / 4-channel full-resolution intermediates.
//
// A wave owns 16 pixels per step; lane (r = l & 15, g = l >> 4) holds for pixel r the channels 16 q + 4 g + e (e = 0..3) of every K group q:
// exactly the B operand of v_mfma_f32_16x16x4_f32 for "step (q, e)", with the weights pre-packed to the same K permutation
// (packing.bins_tail_weights: [nq][5 fragments][64 lanes][4]) and resident in LDS.  The operand is BUILT in registers straight from
// global memory -- K groups 0-1 = `last` (16-byte loads from the CLB buffer), 2-9 = the embedding, blended from its four low-resolution taps
// with the resize kernels' own expression (bit-identical to pf_resize_bilinear), 10 = rel -- so the up-sampled embedding never exists.  The
// accumulators come out as 20 of the 80 hidden channels of the lane's pixel: bias + GELU in place, the 80 -> 4 layer as 80 FMAs per lane + two
// xor-shuffles across the four lane groups, Softplus, then the 64 bins are split 16 per lane group for the log-binomial passes.
constexpr int BT_HID_FRAGS = 5;        // 80 hidden channels
constexpr int BT_MAXQ = 11;            // K groups of 16 channels: 32 last + 128 embedding (+ 8 rel)
__global__ __launch_bounds__(256, 2) void bins_tail_kernel(const float* __restrict__ clb, int clb_ld, int rel_off, const float* __restrict__ emb,
                                                           int he, int we, const float* __restrict__ w0f, const float* __restrict__ b0,
                                                           const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ cen,
                                                           int hc, int wc, float* __restrict__ depth, int B, int H, int W, int nq, float min_temp,
                                                           float max_temp, float she, float swe, float shc, float swc) {
  extern __shared__ __attribute__((aligned(16))) char bt_lds[];
  float4* w0s = reinterpret_cast<float4*>(bt_lds);                                         // [nq][5][64]
  float* w2s = reinterpret_cast<float*>(bt_lds + (size_t)nq * BT_HID_FRAGS * 64 * 16);     // [4][80]
  float* logc = w2s + 320;                                                                 // [64]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
  for (int i = tid; i < nq * BT_HID_FRAGS * 64; i += blockDim.x) w0s[i] = reinterpret_cast<const float4*>(w0f)[i];
  for (int i = tid; i < 320; i += blockDim.x) w2s[i] = w2[i];
  const float eps = 1e-7f;
  const float n_ = 63.0f + eps;
  if (tid < 64) {
    const float k_ = (float)tid + eps;
    logc[tid] = n_ * logf(n_) - k_ * logf(k_) - (n_ - k_) * logf(n_ - k_ + eps);
  }
  __syncthreads();
  float bias0[BT_HID_FRAGS][4];
#pragma unroll
  for (int f = 0; f < BT_HID_FRAGS; ++f)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias0[f][e] = b0[16 * f + 4 * g + e];
  const float b2r[4] = {b2[0], b2[1], b2[2], b2[3]};

  const long total = (long)B * H * W;
  const long ngroups = (total + 15) / 16;
  const long gstride = (long)gridDim.x * 4;
  for (long grp = (long)blockIdx.x * 4 + (tid >> 6); grp < ngroups; grp += gstride) {
    const long pix_raw = grp * 16 + r;
    const bool valid = pix_raw < total;
    const long pix = valid ? pix_raw : total - 1;
    const int ox = (int)(pix % W);
    const int oy = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    // ---- hidden layer: 80 x K on the f32 MFMA; the operand of K group q (channels 16 q + 4 g + e of this lane's pixel) is built just before
    // its 20 MFMAs, the four embedding taps of the NEXT group already in flight (one group of look-ahead keeps the registers at ~130)
    const float* cp = clb + pix * clb_ld + 4 * g;
    const Lerp ly = ac_coord(oy, she, he), lx = ac_coord(ox, swe, we);
    const float* e00 = emb + (((long)b * he + ly.i0) * we + lx.i0) * 128 + 4 * g;
    const float* e01 = emb + (((long)b * he + ly.i0) * we + lx.i1) * 128 + 4 * g;
    const float* e10 = emb + (((long)b * he + ly.i1) * we + lx.i0) * 128 + 4 * g;
    const float* e11 = emb + (((long)b * he + ly.i1) * we + lx.i1) * 128 + 4 * g;
    f32x4 acc[BT_HID_FRAGS];
#pragma unroll
    for (int f = 0; f < BT_HID_FRAGS; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 nx0 = *reinterpret_cast<const float4*>(cp), nx1, nx2, nx3;          // group 0 = last[0:16)
#pragma unroll
    for (int q = 0; q < BT_MAXQ; ++q) {
      if (q < nq) {
        float4 x;
        if (q < 2 || q == 10) x = nx0;
        else x = make_float4(bilerp(ly, lx, nx0.x, nx1.x, nx2.x, nx3.x), bilerp(ly, lx, nx0.y, nx1.y, nx2.y, nx3.y),
                             bilerp(ly, lx, nx0.z, nx1.z, nx2.z, nx3.z), bilerp(ly, lx, nx0.w, nx1.w, nx2.w, nx3.w));
        // look-ahead loads of group q + 1
        if (q + 1 == 1) nx0 = *reinterpret_cast<const float4*>(cp + 16);
        else if (q + 1 >= 2 && q + 1 < 10) {
          nx0 = *reinterpret_cast<const float4*>(e00 + 16 * (q - 1));
          nx1 = *reinterpret_cast<const float4*>(e01 + 16 * (q - 1));
          nx2 = *reinterpret_cast<const float4*>(e10 + 16 * (q - 1));
          nx3 = *reinterpret_cast<const float4*>(e11 + 16 * (q - 1));
        } else if (q + 1 == 10 && nq > 10) {
          nx0 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g < 2) nx0 = *reinterpret_cast<const float4*>(clb + pix * clb_ld + rel_off + 4 * g);
        }
        const float xe[4] = {x.x, x.y, x.z, x.w};
        float4 wf[BT_HID_FRAGS];
#pragma unroll
        for (int f = 0; f < BT_HID_FRAGS; ++f) wf[f] = w0s[(q * BT_HID_FRAGS + f) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int f = 0; f < BT_HID_FRAGS; ++f) {
            const float we_ = e == 0 ? wf[f].x : (e == 1 ? wf[f].y : (e == 2 ? wf[f].z : wf[f].w));
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(we_, xe[e], acc[f], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- bias + GELU, 80 -> 4, Softplus
    float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < BT_HID_FRAGS; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = gelu_erf(acc[f][e] + bias0[f][e]);
        const int ch = 16 * f + 4 * g + e;
#pragma unroll
        for (int c = 0; c < 4; ++c) o4[c] += w2s[c * 80 + ch] * t;
      }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o4[c] += __shfl_xor(o4[c], 16, 64);
      o4[c] += __shfl_xor(o4[c], 32, 64);
      o4[c] = softplus20(o4[c] + b2r[c]);
    }
    // ---- log-binomial softmax over 64 bins (dist_layers.py:97-121), 16 bins per lane group
    const float p0 = o4[0] + 1e-4f, p1 = o4[1] + 1e-4f, t0 = o4[2] + 1e-4f, t1 = o4[3] + 1e-4f;
    float p = p0 / (p0 + p1);
    float t = t0 / (t0 + t1);
    t = (max_temp - min_temp) * t + min_temp;
    const float omp = fminf(fmaxf(1.0f - p, 1e-4f), 1.0f);
    p = fminf(fmaxf(p, 1e-4f), 1.0f);
    const float lp = logf(p), lq = logf(omp);
    float yk[16];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = 16 * g + j;
      yk[j] = (logc[k] + (float)k * lp + (float)(63 - k) * lq) / t;
      mx = fmaxf(mx, yk[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const Lerp cy = ac_coord(oy, shc, hc), cx = ac_coord(ox, swc, wc);
    const long cb = (long)b * hc * wc;
    const float* c00 = cen + (cb + (long)cy.i0 * wc + cx.i0) * 64 + 16 * g;
    const float* c01 = cen + (cb + (long)cy.i0 * wc + cx.i1) * 64 + 16 * g;
    const float* c10 = cen + (cb + (long)cy.i1 * wc + cx.i0) * 64 + 16 * g;
    const float* c11 = cen + (cb + (long)cy.i1 * wc + cx.i1) * 64 + 16 * g;
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int j4 = 0; j4 < 16; j4 += 4) {
      float a00[4], a01[4], a10[4], a11[4];
      load4(c00 + j4, a00); load4(c01 + j4, a01); load4(c10 + j4, a10); load4(c11 + j4, a11);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = expf(yk[j4 + e] - mx);
        const float c = cy.l0 * (cx.l0 * a00[e] + cx.l1 * a01[e]) + cy.l1 * (cx.l0 * a10[e] + cx.l1 * a11[e]);
        num += pe * c;
        den += pe;
      }
    }
    num += __shfl_xor(num, 16, 64); num += __shfl_xor(num, 32, 64);
    den += __shfl_xor(den, 16, 64); den += __shfl_xor(den, 32, 64);
    if (valid && g == 0) depth[pix] = num / den;
  }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define LAUNCH_T(kern, total, ...)                                                                                      \
  do {                                                                                                                  \
    if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(kern<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL(kern<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), __VA_ARGS__);            \
  } while (0)

// one block row per (batch, output row); x covers the row's (column, channel-vector) items
#define LAUNCH_ROWS(kern, rows, items, ...)                                                                             \
  do {                                                                                                                  \
    const dim3 g((unsigned)(((items) + 255) / 256 > 64 ? 64 : ((items) + 255) / 256), (unsigned)((rows) > 65535 ? 65535 : (rows))); \
    if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(kern<bf16_t>, g, dim3(256), 0, ST(stream), __VA_ARGS__);             \
    else hipLaunchKernelGGL(kern<float>, g, dim3(256), 0, ST(stream), __VA_ARGS__);                                     \
  } while (0)

// PF_RESIZE_V2=0: the output-walking kernels (A/B measurements)
static bool resize_v2() {
  const char* e = getenv("PF_RESIZE_V2");      // read per call: the parity test flips it between two launches
  return !(e && e[0] == '0');
}

// grid of the source-aligned kernels: x = up to 4 blocks per source row, y = groups of rpb consecutive source rows of one image;
// LDS = row ranges [8] + column ranges [Wmax] + weights [OW]
constexpr int PF_RESIZE_V1 = -1;      // launch_resize_src: the source-aligned kernel's tables do not fit (W + OW above ~7.4k) -> use the v1 kernel
template <typename T>
static int launch_resize_src(const ResizeSrc* s, int nsrc, int B, void* y, int y_ld, int OH, int OW, const void* add, int add_ld, hipStream_t st) {
  constexpr int N = 16 / sizeof(T);
  int Hmax = 0, Wmax = 0;
  long items = 0;
  for (int i = 0; i < nsrc; ++i) {
    Hmax = s[i].H > Hmax ? s[i].H : Hmax;
    Wmax = s[i].W > Wmax ? s[i].W : Wmax;
    const long it = (long)s[i].W * (s[i].C / N);
    items = it > items ? it : items;
  }
  const size_t lds = RESIZE_RPB * 8 + (size_t)Wmax * 8 + (size_t)OW * 8;
  if (items >= (1L << 22) || lds > 60000) return PF_RESIZE_V1;       // (very wide maps: the caller falls through to the output-walking kernels)
  // rows per block: 8 when that still leaves >= 4 blocks per CU, fewer for small maps
  const long gx = (items + 255) / 256 > 4 ? 4 : (items + 255) / 256;
  int rpb = RESIZE_RPB;
  while (rpb > 1 && (long)B * ((Hmax + rpb - 1) / rpb) * gx < 1024) rpb >>= 1;
  long gy = (long)B * ((Hmax + rpb - 1) / rpb);
  gy = gy > 65535 ? 65535 : gy;
  const ResizeSrc none{};
  hipLaunchKernelGGL(resize_src_kernel<T>, dim3((unsigned)gx, (unsigned)gy), dim3(256), lds, st, s[0], nsrc > 1 ? s[1] : none, nsrc > 2 ? s[2] : none,
                     nsrc, B, Hmax, y, y_ld, OH, OW, add, add_ld, Wmax, rpb);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

extern "C" int pf_resize_bilinear(const void* x, int x_ld, int B, int H, int W, int C, void* y, int y_ld, int OH, int OW,
                                  const void* add, int add_ld, int in_f32, int out_f32, int dtype, void* stream) {
  if (!x || !y || C % 8 || x_ld % 8 || y_ld % 8 || (add && add_ld % 8)) return PF_ERR_ARG;
  const bool homogeneous = dtype == PF_DTYPE_BF16 ? (!in_f32 && !out_f32) : true;      // (dtype f32: everything is float32)
  if (homogeneous && resize_v2()) {
    const ResizeSrc s0{x, x_ld, H, W, C, ac_scale(H, OH), ac_scale(W, OW)};
    const int rc = dtype == PF_DTYPE_BF16 ? launch_resize_src<bf16_t>(&s0, 1, B, y, y_ld, OH, OW, add, add_ld, ST(stream))
                                          : launch_resize_src<float>(&s0, 1, B, y, y_ld, OH, OW, add, add_ld, ST(stream));
    if (rc != PF_RESIZE_V1) return rc;
  }
  LAUNCH_ROWS(resize_bilinear_kernel, B * OH, OW * (C / 8), x, x_ld, B, H, W, C, y, y_ld, OH, OW, add, add_ld, in_f32, out_f32,
              ac_scale(H, OH), ac_scale(W, OW));
  return ok();
}

extern "C" int pf_resize_concat(const void* const* xs, const int* lds, const int* Hs, const int* Ws, const int* Cs, int nsrc, int B,
                                void* y, int y_ld, int OH, int OW, int dtype, void* stream) {
  if (!xs || !lds || !Hs || !Ws || !Cs || !y || nsrc < 2 || nsrc > 3 || y_ld % 8) return PF_ERR_ARG;
  ResizeSrc s[3] = {};
  long cv = 0;
  for (int i = 0; i < nsrc; ++i) {
    if (!xs[i] || Cs[i] % 8 || lds[i] % 8) return PF_ERR_ARG;
    s[i] = ResizeSrc{xs[i], lds[i], Hs[i], Ws[i], Cs[i], ac_scale(Hs[i], OH), ac_scale(Ws[i], OW)};
    cv += Cs[i] / 8;
  }
  int cvmax = 0;
  for (int i = 0; i < nsrc; ++i) cvmax = Cs[i] / 8 > cvmax ? Cs[i] / 8 : cvmax;
  (void)cv;
  if (resize_v2()) {
    const int rc = dtype == PF_DTYPE_BF16 ? launch_resize_src<bf16_t>(s, nsrc, B, y, y_ld, OH, OW, nullptr, 0, ST(stream))
                                          : launch_resize_src<float>(s, nsrc, B, y, y_ld, OH, OW, nullptr, 0, ST(stream));
    if (rc != PF_RESIZE_V1) return rc;
  }
  LAUNCH_ROWS(resize_concat_kernel, B * OH, OW * cvmax, s[0], s[1], s[2], nsrc, B, y, y_ld, OH, OW);
  return ok();
}

extern "C" int pf_crop_resize_planar(const float* img, int C, int H, int W, const int* boxes, int P, float* out, int oh, int ow, void* stream) {
  if (!img || !boxes || !out) return PF_ERR_ARG;
  const long total = (long)P * C * oh * ow;
  hipLaunchKernelGGL(crop_resize_planar_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), img, C, H, W, boxes, P, out, oh, ow);
  return ok();
}

extern "C" int pf_roi_align(const void* feat, int f_ld, int Bf, int H, int W, int C, const float* rois, int K, void* y, int y_ld,
                            int oh, int ow, float spatial_scale, int in_f32, int out_f32, int dtype, void* stream) {
  if (!feat || !rois || !y) return PF_ERR_ARG;
  if (C == 1) {  // planar float depth map
    if (!in_f32 || !out_f32) return PF_ERR_ARG;
    hipLaunchKernelGGL(roi_align_scalar_kernel, dim3(grid_for((long)K * oh * ow, 256)), dim3(256), 0, ST(stream), (const float*)feat, Bf, H, W, rois, K, (float*)y, oh, ow, spatial_scale);
    return ok();
  }
  if (C % 8 || f_ld % 8 || y_ld % 8) return PF_ERR_ARG;
  if ((long)K * oh >= (1L << 31) || (long)(ow / 4 + 1) * (C / 8) >= (1L << 31)) return PF_ERR_ARG;
  const unsigned rows = (unsigned)((long)K * oh);
  if (dtype == PF_DTYPE_BF16)
    hipLaunchKernelGGL(roi_align_kernel<bf16_t>, dim3(rows), dim3(256), 0, ST(stream), feat, f_ld, Bf, H, W, C, rois, K, y, y_ld, oh, ow, spatial_scale, in_f32, out_f32);
  else
    hipLaunchKernelGGL(roi_align_kernel<float>, dim3(rows), dim3(256), 0, ST(stream), feat, f_ld, Bf, H, W, C, rois, K, y, y_ld, oh, ow, spatial_scale, in_f32, out_f32);
  return ok();
}

extern "C" int pf_maxpool2(const void* x, int x_ld, int B, int H, int W, int C, void* y, int y_ld, int dtype, void* stream) {
  if (!x || !y || C % 8 || x_ld % 8 || y_ld % 8) return PF_ERR_ARG;
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(maxpool2_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (const bf16_t*)x, x_ld, B, H, W, C, (bf16_t*)y, y_ld);
  else hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (const float*)x, x_ld, B, H, W, C, (float*)y, y_ld);
  return ok();
}

extern "C" int pf_copy_channels(const void* x, int x_ld, void* y, int y_ld, long npix, int C, int in_f32, int out_f32, int dtype, void* stream) {
  if (!x || !y || C % 8 || x_ld % 8 || y_ld % 8) return PF_ERR_ARG;
  const long total = npix * (C / 8);
  LAUNCH_T(copy_channels_kernel, total, x, x_ld, y, y_ld, npix, C, in_f32, out_f32);
  return ok();
}

extern "C" int pf_pack_fusion_input(const float* cdepth, const float* fdepth, const float* crops, void* y, int B, int h, int w, int dtype, void* stream) {
  if (!cdepth || !fdepth || !crops || !y) return PF_ERR_ARG;
  const long total = (long)B * h * w;
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(pack_fusion_input_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), cdepth, fdepth, crops, (bf16_t*)y, B, h, w);
  else hipLaunchKernelGGL(pack_fusion_input_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), cdepth, fdepth, crops, (float*)y, B, h, w);
  return ok();
}

extern "C" int pf_nhwc_to_nchw_f32(const void* x, int x_ld, float* y, int B, int H, int W, int C, int in_f32, int dtype, void* stream) {
  if (!x || !y) return PF_ERR_ARG;
  const long total = (long)B * C * H * W;
  LAUNCH_T(nhwc_to_nchw_kernel, total, x, x_ld, y, B, H, W, C, in_f32);
  return ok();
}

extern "C" int pf_attractor(const float* A, int a_ld, int n_attr, int a_stride, float a_eps, int attractor_exp, int kind_sum,
                            const float* b_prev, int hp, int wp, float* out, int B, int h, int w, int n_bins, void* stream) {
  if (!A || !b_prev || !out || n_bins % 4 || n_attr < 1 || a_stride < 1) return PF_ERR_ARG;
  const long total = (long)B * h * w * (n_bins / 4);
  const float scale = kind_sum ? 1.0f : 1.0f / (float)n_attr;
  if (attractor_exp)
    hipLaunchKernelGGL(attractor_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), A, a_ld, n_attr, a_stride, a_eps, scale,
                       b_prev, hp, wp, out, B, h, w, n_bins, ac_scale(hp, h), ac_scale(wp, w));
  else
    hipLaunchKernelGGL(attractor_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), A, a_ld, n_attr, a_stride, a_eps, scale,
                       b_prev, hp, wp, out, B, h, w, n_bins, ac_scale(hp, h), ac_scale(wp, w));
  return ok();
}

extern "C" int pf_seed_bin_centers(const float* x, int x_ld, float* out, long npix, int n_bins, float min_depth, float max_depth,
                                   int bounded, int normalize, void* stream) {
  if (!x || !out || npix <= 0 || n_bins < 1 || x_ld < n_bins) return PF_ERR_ARG;
  hipLaunchKernelGGL(seed_bin_centers_kernel, dim3(grid_for(npix, 64)), dim3(64), 0, ST(stream), x, x_ld, out, npix, n_bins, min_depth,
                     max_depth, bounded, normalize);
  return ok();
}

extern "C" int pf_bounded_bin_centers(const float* b, float* out, long npix, int n_bins, float min_depth, float max_depth, void* stream) {
  if (!b || !out || npix <= 0 || n_bins < 1 || n_bins > 64) return PF_ERR_ARG;
  hipLaunchKernelGGL(bounded_centers_kernel, dim3(grid_for(npix * 64, 256)), dim3(256), 0, ST(stream), b, out, npix, n_bins, min_depth,
                     max_depth);
  return ok();
}

extern "C" int pf_logbinom_depth(const float* pt, int pt_ld, const float* centers, int hc, int wc, float* depth, int B, int h, int w,
                                 int n_bins, float min_temp, float max_temp, void* stream) {
  if (!pt || !centers || !depth || n_bins % 4 || n_bins > 256) return PF_ERR_ARG;
  const long total = (long)B * h * w;
  hipLaunchKernelGGL(logbinom_depth_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), pt, pt_ld, centers, hc, wc, depth, B, h, w, n_bins, min_temp, max_temp, ac_scale(hc, h), ac_scale(wc, w));
  return ok();
}

extern "C" int pf_stitch_init(float* pred, float* count, int MH, int MW, const float* depth, const float* mask, const int* yx, int P,
                              int ph, int pw, void* stream) {
  if (!pred || !count || !depth || !mask || !yx) return PF_ERR_ARG;
  const long total = (long)P * ph * pw;
  hipLaunchKernelGGL(stitch_init_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), pred, count, MH, MW, depth, mask, yx, P, ph, pw);
  return ok();
}
extern "C" int pf_stitch_finish_init(float* avg, const float* pred, const float* count, long n, void* stream) {
  if (!avg || !pred || !count) return PF_ERR_ARG;
  hipLaunchKernelGGL(div_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(stream), avg, pred, count, n);
  return ok();
}
extern "C" int pf_stitch_update(float* avg, float* count, int MH, int MW, const float* depth, int dh, int dw, const float* mask, int y0,
                                int x0, int ph, int pw, void* stream) {
  if (!avg || !count || !depth || !mask) return PF_ERR_ARG;
  const long total = (long)ph * pw;
  hipLaunchKernelGGL(stitch_update_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), avg, count, MH, MW, depth, dh, dw, mask, y0, x0, ph, pw, (float)dh / (float)ph, (float)dw / (float)pw);
  return ok();
}
extern "C" int pf_resize_nearest_f32(const float* x, int H, int W, float* y, int OH, int OW, void* stream) {
  if (!x || !y) return PF_ERR_ARG;
  const long total = (long)OH * OW;
  hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), x, H, W, y, OH, OW, (float)H / (float)OH, (float)W / (float)OW);
  return ok();
}
extern "C" int pf_resize_bilinear_f32(const float* x, int H, int W, float* y, int OH, int OW, void* stream) {
  if (!x || !y) return PF_ERR_ARG;
  const long total = (long)OH * OW;
  hipLaunchKernelGGL(resize_bilinear_f32_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), x, H, W, y, OH, OW, ac_scale(H, OH), ac_scale(W, OW));
  return ok();
}

extern "C" int pf_bins_tail(const float* clb, int clb_ld, int rel_off, const float* emb, int he, int we, const float* w0f, const float* b0,
                            const float* w2, const float* b2, const float* centers, int hc, int wc, float* depth, int B, int H, int W, int nq,
                            float min_temp, float max_temp, void* stream) {
  if (!clb || !emb || !w0f || !b0 || !w2 || !b2 || !centers || !depth || clb_ld % 4 || (nq != 10 && nq != 11) || (nq == 11 && (rel_off % 4 || rel_off + 8 > clb_ld)) ||
      clb_ld < 32 || B <= 0 || H <= 0 || W <= 0)
    return PF_ERR_ARG;
  const size_t lds = (size_t)nq * BT_HID_FRAGS * 64 * 16 + 320 * 4 + 64 * 4;
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(bins_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    done.fetch_or(bit, std::memory_order_release);
  }
  const long groups = ((long)B * H * W + 15) / 16;
  long blocks = (groups + 3) / 4;
  blocks = blocks > 512 ? 512 : blocks;                     // two resident blocks per CU, each looping over its share of the pixels
  hipLaunchKernelGGL(bins_tail_kernel, dim3((unsigned)blocks), dim3(256), lds, ST(stream), clb, clb_ld, rel_off, emb, he, we, w0f, b0, w2, b2, centers, hc,
                     wc, depth, B, H, W, nq, min_temp, max_temp, ac_scale(he, H), ac_scale(we, W), ac_scale(hc, H), ac_scale(wc, W));
  return ok();
}
