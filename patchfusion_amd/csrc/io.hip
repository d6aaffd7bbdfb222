// Input / output side of the tiled-inference path: everything the reference does
// on the CPU between the decoded uint8 image and `image_hr`/`image_lr`, and between the stitched depth map and
// the files / metrics it writes.  All of it is HBM-bound byte/float streaming; nothing here is GEMM shaped.
//   * u8_bicubic_kernel     - estimator/datasets/general_dataset.py:22-47 (`img / 255.0`, float64 bicubic
//                             align_corners=True to image_raw_shape, `.float()`), channel reversal of the 'u4k' branch
//   * exact percentiles     - np.percentile(value[mask], 2 / 95) of estimator/utils/color.py:127-128 as a three-level
//                             radix select on order-preserving integer keys (bit-exact order statistics), linear
//                             interpolation in double as numpy's _lerp
//   * colorize_kernel       - estimator/utils/color.py:130-150 + matplotlib Colormap.__call__(bytes=True)
//   * depth_u16_kernel      - estimator/tester/tester.py:75 `(depth * 256).astype('uint16')`
//   * depth_metrics_kernel  - estimator/utils/metric.py:10-52 (compute_errors), :66-71 (soft_edge_error), :87-148
//                             (compute_metrics: resize, clamp, masks) as one fused masked reduction
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
inline int ok() { return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH; }
#define ST(s) reinterpret_cast<hipStream_t>(s)

// ------------------------------------------------------------------------------------------------
// uint8 HWC -> float32 CHW in [0,1], bicubic (A = -0.75) with align_corners=True, evaluated in double like the
// reference (numpy `/ 255.0` promotes to float64 before F.interpolate).  torch's separable order: horizontal taps
// first, out = sum_j wy[j] * (sum_i wx[i] * v[j][i]); out-of-range taps clamp to the border.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double cc1(double x) {
  const double A = -0.75;
  return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0;
}
__device__ __forceinline__ double cc2(double x) {
  const double A = -0.75;
  return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A;
}
__device__ __forceinline__ void cubic_coeffs(double t, double (&c)[4]) {
  c[0] = cc2(t + 1.0);
  c[1] = cc1(t);
  const double x2 = 1.0 - t;
  c[2] = cc1(x2);
  c[3] = cc2(x2 + 1.0);
}

struct __attribute__((packed, aligned(1))) Bytes12 {
  uint32_t w[3];
};

__global__ __launch_bounds__(256) void u8_bicubic_kernel(const uint8_t* __restrict__ src, int H, int W, int reverse,
                                                         float* __restrict__ dst, int OH, int OW, double sh, double sw) {
  __shared__ double unit[256];                  // k / 255.0, the reference's first operation, exactly as numpy rounds it
  unit[threadIdx.x] = (double)threadIdx.x / 255.0;
  __syncthreads();
  const long total = (long)OH * OW, plane = total;
  const bool same = (H == OH && W == OW);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int oy = (int)(i / OW), ox = (int)(i - (long)oy * OW);
    double acc[3];
    if (same) {                                 // the cubic weights at t = 0 are exactly {0, 1, 0, 0}
      const uint8_t* px = src + ((long)oy * W + ox) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = unit[px[c]];
    } else {
      const double ry = sh * oy, rx = sw * ox;
      const int iy = (int)floor(ry), ix = (int)floor(rx);
      double cy[4], cx[4];
      cubic_coeffs(ry - iy, cy);
      cubic_coeffs(rx - ix, cx);
      int xs[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) xs[k] = min(max(ix - 1 + k, 0), W - 1) * 3;
      const bool inner = ix >= 1 && ix + 2 <= W - 1;   // the four taps are 12 contiguous bytes: one (unaligned) load
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint8_t* row = src + (long)min(max(iy - 1 + j, 0), H - 1) * W * 3;
        uint8_t tap[12];
        if (inner) {
          const Bytes12 q = *reinterpret_cast<const Bytes12*>(row + xs[0]);
#pragma unroll
          for (int e = 0; e < 12; ++e) tap[e] = (uint8_t)(q.w[e >> 2] >> (8 * (e & 3)));
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) tap[3 * k + c] = row[xs[k] + c];
        }
        double t[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double v = unit[tap[3 * k + c]] * cx[k];
            t[c] = k == 0 ? v : t[c] + v;
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double o = t[c] * cy[j];
          acc[c] = j == 0 ? o : acc[c] + o;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(long)(reverse ? 2 - c : c) * plane + i] = (float)acc[c];
  }
}

// ------------------------------------------------------------------------------------------------
// Exact percentiles by radix select.  key(f) is an order-preserving map float32 -> uint32; three histogram levels
// (11 + 11 + 10 bits) pin the lower and upper order statistic of each of two percentiles (four ranks).
// Workspace (device, PF_PERCENTILE_WS_BYTES): 4 histograms of 2048 bins + the select state.
// ------------------------------------------------------------------------------------------------
constexpr int PCT_R = 4, PCT_BINS = 2048;
struct PctState {
  unsigned long long rank[PCT_R];   // remaining rank inside the current prefix
  unsigned int prefix[PCT_R];
  unsigned long long n;             // number of valid samples
  double gamma[2];                  // fractional part of the virtual index of each percentile
};
struct PctWs {
  unsigned int hist[PCT_R][PCT_BINS];
  PctState st;
};

__device__ __forceinline__ unsigned int f2key(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void pct_init_kernel(PctWs* ws) {
  unsigned int* p = reinterpret_cast<unsigned int*>(ws);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (int)(sizeof(PctWs) / 4); i += gridDim.x * blockDim.x) p[i] = 0u;
}

template <int LEVEL>
__global__ __launch_bounds__(256) void pct_hist_kernel(const float* __restrict__ x, long n, float invalid, int use_invalid,
                                                       const uint8_t* __restrict__ imask, PctWs* ws) {
  constexpr int NH = LEVEL == 0 ? 1 : PCT_R;
  __shared__ unsigned int h[NH][PCT_BINS];
  for (int i = threadIdx.x; i < NH * PCT_BINS; i += blockDim.x) (&h[0][0])[i] = 0u;
  unsigned int pre[PCT_R];
#pragma unroll
  for (int r = 0; r < PCT_R; ++r) pre[r] = LEVEL == 0 ? 0u : ws->st.prefix[r];
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const bool inval = imask ? imask[i] != 0 : (use_invalid && v == invalid);   // color.py:121-122: an explicit mask replaces the value test
    if (LEVEL != 0 && inval) continue;
    const unsigned int k = f2key(v);
    if (LEVEL == 0) {
      // depth maps put most pixels into a handful of (sign, exponent) bins: 64 lanes hammering one LDS counter
      // serialize.  Peel the (up to 8) most common bins of the wave with ballots - one atomic per bin - and let
      // the stragglers add individually.
      const unsigned int bin = k >> 21;
      bool pending = !inval;
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        const unsigned long long live = __ballot(pending);
        if (!live) break;
        const int leader = __ffsll((long long)live) - 1;
        const unsigned int b0 = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(pending && bin == b0);
        if (pending && bin == b0) {
          pending = false;
          if ((threadIdx.x & 63) == leader) atomicAdd(&h[0][b0], (unsigned int)__popcll(same));
        }
      }
      if (pending) atomicAdd(&h[0][bin], 1u);
    } else if (LEVEL == 1) {
#pragma unroll
      for (int r = 0; r < PCT_R; ++r)
        if ((k >> 21) == pre[r]) atomicAdd(&h[r][(k >> 10) & 2047u], 1u);
    } else {
#pragma unroll
      for (int r = 0; r < PCT_R; ++r)
        if ((k >> 10) == pre[r]) atomicAdd(&h[r][k & 1023u], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NH * PCT_BINS; i += blockDim.x) {
    const unsigned int c = (&h[0][0])[i];
    if (c) atomicAdd(&(&ws->hist[0][0])[i], c);
  }
}

// one block; thread r < 4 walks the histogram of rank r.  After level 2 the four order statistics are known and the
// two percentiles are interpolated exactly like numpy 1.24's _quantile/_lerp (the version the reference pins):
// virtual index (n-1)*q/100 and gamma in double, (b - a) in float32, a + diff*gamma in double, and
// b - diff*(1-gamma) instead when gamma >= 0.5.
template <int LEVEL>
__global__ __launch_bounds__(256) void pct_select_kernel(PctWs* ws, double q0, double q1, float* out) {
  constexpr int bins = LEVEL == 2 ? 1024 : 2048, NH = LEVEL == 0 ? 1 : PCT_R, PER = bins / 256;
  __shared__ unsigned int h[NH][bins];
  __shared__ unsigned int part[NH][256];         // sums of PER consecutive bins
  const int tid = threadIdx.x;
  for (int i = tid; i < NH * PCT_BINS; i += 256) {
    const int r = i / PCT_BINS, b = i - r * PCT_BINS;
    if (b < bins) h[r][b] = ws->hist[r][b];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NH; ++r) {
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) c += h[r][tid * PER + j];
    part[r][tid] = c;
  }
  __syncthreads();
  if (LEVEL == 0 && tid == 0) {
    unsigned long long n = 0;
    for (int i = 0; i < 256; ++i) n += part[0][i];
    ws->st.n = n;
    const double q[2] = {q0, q1};
    for (int p = 0; p < 2; ++p) {
      const double vi = n ? (double)(n - 1) * (q[p] / 100.0) : 0.0;
      const double lo = floor(vi);
      unsigned long long l = (unsigned long long)lo, hgh = l + 1;
      if (n && hgh > n - 1) hgh = n - 1;
      ws->st.rank[2 * p] = l;
      ws->st.rank[2 * p + 1] = n ? hgh : 0;
      ws->st.gamma[p] = vi - lo;
    }
  }
  __syncthreads();
  if (tid < PCT_R) {
    const int r = tid, hr = LEVEL == 0 ? 0 : r;
    unsigned long long rank = ws->st.rank[r], cum = 0;
    int g = 0;
    for (; g < 255; ++g) {                        // group of PER bins that holds the rank
      const unsigned long long c = part[hr][g];
      if (rank < cum + c) break;
      cum += c;
    }
    int b = g * PER;
    for (; b < g * PER + PER - 1; ++b) {
      const unsigned long long c = h[hr][b];
      if (rank < cum + c) break;
      cum += c;
    }
    ws->st.rank[r] = rank - cum;
    ws->st.prefix[r] = LEVEL == 0 ? (unsigned int)b : ((ws->st.prefix[r] << (LEVEL == 2 ? 10 : 11)) | (unsigned int)b);
  }
  __syncthreads();
  if (LEVEL < 2) {
    for (int i = tid; i < PCT_R * PCT_BINS; i += 256) (&ws->hist[0][0])[i] = 0u;
  } else if (tid < 2) {
    const int p = tid;
    if (ws->st.n == 0) {
      out[p] = __uint_as_float(0x7fc00000u);
    } else {
      const float a = key2f(ws->st.prefix[2 * p]), b = key2f(ws->st.prefix[2 * p + 1]);
      const float diff = b - a;
      const double t = ws->st.gamma[p];
      double v = (double)a + (double)diff * t;
      if (t >= 0.5) v = (double)b - (double)diff * (1.0 - t);
      out[p] = (float)v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// colorize: x = (v - vmin) / (vmax - vmin) in float32 (v * 0 when vmin == vmax), invalid -> NaN, then matplotlib's
// Colormap.__call__(x, bytes=True): x*N, <0 -> under, ==N -> N-1, >N-1 -> over, NaN -> bad; lut has N+3 RGBA rows
// (N colours, under, over, bad) already scaled to bytes by the host; invalid pixels get the background colour.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colorize_kernel(const float* __restrict__ depth, long n, const float* __restrict__ vmm,
                                                       const uint32_t* __restrict__ lut, int N, float invalid, int use_invalid,
                                                       const uint8_t* __restrict__ imask, uint32_t background, uint32_t* __restrict__ out) {
  const float vmin = vmm[0], vmax = vmm[1];
  const float den = vmax - vmin;
  const float fN = (float)N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = depth[i];
    if (imask ? imask[i] != 0 : (use_invalid && v == invalid)) {
      out[i] = background;
      continue;
    }
    float x = vmin != vmax ? __fdiv_rn(v - vmin, den) : v * 0.f;
    int idx;
    if (x != x) {
      idx = N + 2;
    } else {
      x = x * fN;
      if (x < 0.f) idx = N;             // under
      else if (x == fN) idx = N - 1;
      else if (x > fN) idx = N + 1;     // over (clip(-1, N) then astype(int) > N-1)
      else idx = (int)x;
    }
    out[i] = lut[idx];
  }
}

__global__ __launch_bounds__(256) void depth_u16_kernel(const float* __restrict__ depth, long n, float scale, uint16_t* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = depth[i] * scale;                           // float32 product, then C truncation like ndarray.astype
    out[i] = (uint16_t)(v >= 65535.f ? 65535 : (v > 0.f ? (int)v : 0));
  }
}

// ------------------------------------------------------------------------------------------------
// depth metrics: one pass over the ground-truth grid.  pred is resized on the fly (bilinear, align_corners=False,
// float32) when its grid differs, clamped to [min,max] (inf -> max, nan -> min), masked by min < gt < max and the
// evaluation rectangle; every term is computed in float32 as numpy does and accumulated in double.
// out[0..12] = n, #(thresh<1.25), #(<1.25^2), #(<1.25^3), sum|gt-p|/gt, sum (gt-p)^2/gt, sum (gt-p)^2,
//              sum (log gt - log p)^2, sum (log p - log gt), sum (log p - log gt)^2, sum |log10 gt - log10 p|,
//              sum soft-edge error over (mask & edges), #(mask & edges)
// ------------------------------------------------------------------------------------------------
constexpr int MET_N = 13;

__device__ __forceinline__ float pred_at(const float* __restrict__ pred, int ph, int pw, int H, int W, int y, int x, float sy, float sx,
                                         float lo, float hi) {
  float v;
  if (ph == H && pw == W) {
    v = pred[(long)y * W + x];
  } else {
    float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < ph - 1 ? 1 : 0), x1 = x0 + (x0 < pw - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float t0 = pred[(long)y0 * pw + x0] * hx + pred[(long)y0 * pw + x1] * lx;
    const float t1 = pred[(long)y1 * pw + x0] * hx + pred[(long)y1 * pw + x1] * lx;
    v = t0 * hy + t1 * ly;
  }
  if (v != v) return lo;               // nan -> min (after the < / > clamps, which leave nan alone)
  if (v < lo) v = lo;
  if (v > hi) v = hi;                  // also +inf -> max
  return v;
}

__global__ __launch_bounds__(256) void depth_metrics_kernel(const float* __restrict__ gt, int H, int W, const float* __restrict__ pred,
                                                            int ph, int pw, const float* __restrict__ edges,
                                                            const uint8_t* __restrict__ extra, float lo, float hi, int cy0, int cy1, int cx0,
                                                            int cx1, double* __restrict__ out) {
  double s[MET_N];
#pragma unroll
  for (int k = 0; k < MET_N; ++k) s[k] = 0.0;
  const float sy = (float)ph / (float)H, sx = (float)pw / (float)W;
  const long total = (long)H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    const float g = gt[i];
    if (!(g > lo && g < hi) || y < cy0 || y >= cy1 || x < cx0 || x >= cx1) continue;
    if (extra && extra[i] == 0) continue;                 // metric.py:128-130 additional_mask (prompt-depth evaluation)
    const float p = pred_at(pred, ph, pw, H, W, y, x, sy, sx, lo, hi);
    const float th = fmaxf(g / p, p / g);
    const float d = g - p;
    const float lg = logf(g), lp = logf(p);
    const float e = lp - lg;
    s[0] += 1.0;
    s[1] += th < 1.25f ? 1.0 : 0.0;
    s[2] += th < 1.5625f ? 1.0 : 0.0;
    s[3] += th < 1.953125f ? 1.0 : 0.0;
    s[4] += (double)(fabsf(d) / g);
    s[5] += (double)((d * d) / g);
    s[6] += (double)(d * d);
    s[7] += (double)((lg - lp) * (lg - lp));
    s[8] += (double)e;
    s[9] += (double)(e * e);
    s[10] += (double)fabsf(log10f(g) - log10f(p));
    if (edges && edges[i] != 0.f) {
      float m = 3.4e38f;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y - dy, xx = x - dx;          // shift_2d_replace: shifted[y][x] = gt[y-dy][x-dx], 0 outside
          const float gs = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? gt[(long)yy * W + xx] : 0.f;
          m = fminf(m, fabsf(gs - p));
        }
      s[11] += (double)m;
      s[12] += 1.0;
    }
  }
  __shared__ double red[MET_N][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < MET_N; ++k) {
    double v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < MET_N) {
    const double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (v != 0.0) atomicAdd(&out[threadIdx.x], v);
  }
}

// ------------------------------------------------------------------------------------------------
// SILogLoss forward (estimator/models/losses.py:15-62, the loss of PatchFusion.forward(mode='train'), patchfusion.py:395):
// mask = min < target < max; g = log(input + 1e-7) - log(target + 1e-7) in float32 like torch; the masked count, sum and
// sum of squares are accumulated in double; loss = 10 * sqrt(var_unbiased(g) + beta * mean(g)^2); 0 when <= 1 valid pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silog_sums_kernel(const float* __restrict__ pred, const float* __restrict__ target, long n,
                                                         float lo, float hi, double* __restrict__ out3) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float t = target[i];
    if (!(t > lo && t < hi)) continue;
    const float g = logf(pred[i] + 1e-7f) - logf(t + 1e-7f);
    s0 += 1.0;
    s1 += (double)g;
    s2 += (double)g * (double)g;
  }
  __shared__ double red[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[3] = {s0, s1, s2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane == 0) red[k][wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double r = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (r != 0.0) atomicAdd(&out3[threadIdx.x], r);
  }
}
__global__ void silog_finish_kernel(const double* __restrict__ s, float beta, float* __restrict__ loss) {
  if (threadIdx.x != 0) return;
  const double n = s[0];
  if (n <= 1.0) { *loss = 0.f; return; }
  const double mean = s[1] / n;
  const double var = fmax((s[2] - s[1] * s[1] / n) / (n - 1.0), 0.0);
  *loss = (float)(10.0 * sqrt(var + (double)beta * mean * mean));
}

__global__ void zero_f64_kernel(double* p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0.0;
}

}  // namespace

extern "C" int pf_u8_bicubic_to_f32(const uint8_t* src, int H, int W, int reverse_channels, float* dst, int OH, int OW, void* stream) {
  if (!src || !dst || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return PF_ERR_ARG;
  const double sh = OH > 1 ? (double)(H - 1) / (double)(OH - 1) : 0.0;
  const double sw = OW > 1 ? (double)(W - 1) / (double)(OW - 1) : 0.0;
  hipLaunchKernelGGL(u8_bicubic_kernel, dim3(grid_for((long)OH * OW, 256)), dim3(256), 0, ST(stream), src, H, W, reverse_channels, dst, OH,
                     OW, sh, sw);
  return ok();
}

extern "C" int pf_percentile_workspace_bytes(void) { return (int)sizeof(PctWs); }

extern "C" int pf_percentiles_f32(const float* x, long n, float invalid_val, int use_invalid, const uint8_t* invalid_mask, double q0,
                                  double q1, float* out2, void* workspace, void* stream) {
  if (!x || !out2 || !workspace || n <= 0) return PF_ERR_ARG;
  if (!(q0 >= 0.0 && q0 <= 100.0 && q1 >= 0.0 && q1 <= 100.0)) return PF_ERR_ARG;
  PctWs* ws = reinterpret_cast<PctWs*>(workspace);
  hipStream_t st = ST(stream);
  const int g = grid_for(n, 256 * 8);
  hipLaunchKernelGGL(pct_init_kernel, dim3(8), dim3(256), 0, st, ws);
  hipLaunchKernelGGL(pct_hist_kernel<0>, dim3(g), dim3(256), 0, st, x, n, invalid_val, use_invalid, invalid_mask, ws);
  hipLaunchKernelGGL(pct_select_kernel<0>, dim3(1), dim3(256), 0, st, ws, q0, q1, out2);
  hipLaunchKernelGGL(pct_hist_kernel<1>, dim3(g), dim3(256), 0, st, x, n, invalid_val, use_invalid, invalid_mask, ws);
  hipLaunchKernelGGL(pct_select_kernel<1>, dim3(1), dim3(256), 0, st, ws, q0, q1, out2);
  hipLaunchKernelGGL(pct_hist_kernel<2>, dim3(g), dim3(256), 0, st, x, n, invalid_val, use_invalid, invalid_mask, ws);
  hipLaunchKernelGGL(pct_select_kernel<2>, dim3(1), dim3(256), 0, st, ws, q0, q1, out2);
  return ok();
}

extern "C" int pf_colorize_f32(const float* depth, long n, const float* vmin_vmax, const uint8_t* lut_rgba, int N, float invalid_val,
                               int use_invalid, const uint8_t* invalid_mask, uint32_t background_rgba, uint8_t* out_rgba, void* stream) {
  if (!depth || !vmin_vmax || !lut_rgba || !out_rgba || n <= 0 || N <= 0) return PF_ERR_ARG;
  hipLaunchKernelGGL(colorize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(stream), depth, n, vmin_vmax,
                     reinterpret_cast<const uint32_t*>(lut_rgba), N, invalid_val, use_invalid, invalid_mask, background_rgba,
                     reinterpret_cast<uint32_t*>(out_rgba));
  return ok();
}

extern "C" int pf_depth_to_u16(const float* depth, long n, float scale, uint16_t* out, void* stream) {
  if (!depth || !out || n <= 0) return PF_ERR_ARG;
  hipLaunchKernelGGL(depth_u16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(stream), depth, n, scale, out);
  return ok();
}

extern "C" int pf_depth_metrics(const float* gt, int H, int W, const float* pred, int ph, int pw, const float* edges,
                                const uint8_t* additional_mask, float min_depth, float max_depth, int crop_y0, int crop_y1, int crop_x0,
                                int crop_x1, double* out13, void* stream) {
  if (!gt || !pred || !out13 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0) return PF_ERR_ARG;
  hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(64), 0, ST(stream), out13, MET_N);
  hipLaunchKernelGGL(depth_metrics_kernel, dim3(grid_for((long)H * W, 256)), dim3(256), 0, ST(stream), gt, H, W, pred, ph, pw, edges,
                     additional_mask, min_depth, max_depth, crop_y0, crop_y1, crop_x0, crop_x1, out13);
  return ok();
}

extern "C" int pf_silog_loss(const float* pred, const float* target, long n, float min_depth, float max_depth, float beta, double* ws3,
                             float* loss, void* stream) {
  if (!pred || !target || !ws3 || !loss || n <= 0) return PF_ERR_ARG;
  hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(64), 0, ST(stream), ws3, 3);
  hipLaunchKernelGGL(silog_sums_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ST(stream), pred, target, n, min_depth, max_depth, ws3);
  hipLaunchKernelGGL(silog_finish_kernel, dim3(1), dim3(64), 0, ST(stream), ws3, beta, loss);
  return ok();
}
