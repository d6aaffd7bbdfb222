// G2L (Swin) kernels: LayerNorm + pad + cyclic shift + window partition, the 144-token window
// attention with relative-position bias and shift mask, and window reverse + residual.
// head_dim is tiny here (2..32) -- MFMA-hostile -- so the attention runs on the VALU with K/V of one
// (window, head) in LDS (broadcast reads) and one query per thread; it is HBM/LDS bound.
#include "pf_common.h"
#include <atomic>
#include <cstdlib>
#include "../../include/pf_hip.h"

namespace {

constexpr int WIN = 12, WTOK = 144;

// group of G = C/8 lanes per token; each lane owns 8 channels
template <typename T>
__global__ __launch_bounds__(256) void swin_ln_partition_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ xw,
                                                                const float* __restrict__ gam, const float* __restrict__ bet,
                                                                float eps, int B, int H, int W, int C, int shift) {
  const int G = C >> 3;  // lanes per token (4, 8, 16, 32)
  const int Hp = (H + WIN - 1) / WIN * WIN, Wp = (W + WIN - 1) / WIN * WIN;
  const int nwx = Wp / WIN, nwy = Hp / WIN;
  const long ntok = (long)B * nwy * nwx * WTOK;
  const int tpb = 256 / G;
  const int sub = threadIdx.x % G;
  for (long tok = (long)blockIdx.x * tpb + threadIdx.x / G; tok < ntok; tok += (long)gridDim.x * tpb) {
    const int p = (int)(tok % WTOK);
    long wi = tok / WTOK;
    const int wx = (int)(wi % nwx);
    wi /= nwx;
    const int wy = (int)(wi % nwy);
    const int b = (int)(wi / nwy);
    const int sy = (wy * WIN + p / WIN + shift) % Hp, sx = (wx * WIN + p % WIN + shift) % Wp;
    const bool real = sy < H && sx < W;
    float v[8];
    if (real) load8(x + (((long)b * H + sy) * W + sx) * x_ld + sub * 8, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
    for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / C + eps);
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = real ? (v[e] - mean) * rstd * gam[sub * 8 + e] + bet[sub * 8 + e] : 0.f;
    store8(xw + tok * C + sub * 8, o8);
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(192) void swin_window_attention_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                    const float* __restrict__ bias_table, int Hp, int Wp,
                                                                    int C, int heads, int shift) {
  __shared__ float Ks[WTOK][HD];
  __shared__ float Vs[WTOK][HD];
  __shared__ float bias[529];
  __shared__ int region[WTOK];
  const int win = blockIdx.x, head = blockIdx.y;
  const int nwx = Wp / WIN, nwy = Hp / WIN;
  const int wloc = win % (nwx * nwy);
  const int wy = wloc / nwx, wx = wloc % nwx;
  const int t = threadIdx.x;
  const long row0 = (long)win * WTOK;
  for (int i = t; i < 529; i += 192) bias[i] = bias_table[i * heads + head];
  float qv[HD];
  if (t < WTOK) {
    const T* base = qkv + (row0 + t) * 3 * C + head * HD;
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      qv[d] = Elem<T>::ld(base + d) * scale;
      Ks[t][d] = Elem<T>::ld(base + C + d);
      Vs[t][d] = Elem<T>::ld(base + 2 * C + d);
    }
    const int sy = wy * WIN + t / WIN, sx = wx * WIN + t % WIN;
    const int hid = sy < Hp - WIN ? 0 : (sy < Hp - shift ? 1 : 2);
    const int wid = sx < Wp - WIN ? 0 : (sx < Wp - shift ? 1 : 2);
    region[t] = shift > 0 ? hid * 3 + wid : 0;
  }
  __syncthreads();
  if (t >= WTOK) return;
  const int yi = t / WIN, xi = t % WIN;
  const int bi = (yi + WIN - 1) * (2 * WIN - 1) + xi + WIN - 1;  // index(i,j) = bi - (yj*23 + xj)
  const int myreg = region[t];
  // pass 1: row max
  float mx = -INFINITY;
  for (int jy = 0; jy < WIN; ++jy)
    for (int jx = 0; jx < WIN; ++jx) {
      const int jt = jy * WIN + jx;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s += qv[d] * Ks[jt][d];
      s += bias[bi - (jy * (2 * WIN - 1) + jx)];
      if (region[jt] != myreg) s += -100.0f;
      mx = fmaxf(mx, s);
    }
  // pass 2: softmax-weighted sum
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float l = 0.f;
  for (int jy = 0; jy < WIN; ++jy)
    for (int jx = 0; jx < WIN; ++jx) {
      const int jt = jy * WIN + jx;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s += qv[d] * Ks[jt][d];
      s += bias[bi - (jy * (2 * WIN - 1) + jx)];
      if (region[jt] != myreg) s += -100.0f;
      const float pe = expf(s - mx);
      l += pe;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] += pe * Vs[jt][d];
    }
  const float inv = 1.0f / l;
  T* dst = out + (row0 + t) * C + head * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) Elem<T>::st(dst + d, acc[d] * inv);
}


// Window attention on the f32 matrix pipe (round 6).  A block owns one window and one 32-channel slab of
// the qkv rows (32 / HD heads): q (pre-scaled), k, v of the slab are staged with whole-line reads into LDS
// rows of 36 floats (the 16 x 4 fragment reads of v_mfma_f32_16x16x4_f32 then touch 64 distinct banks).  A
// wave takes (16 queries, head) items: S^T = K Q^T as nine 16-key tiles (so that a lane keeps ONE query's
// scores: keys 16 kt + 4 (lane / 16) + i, query lane % 16), bias / shift mask / softmax in registers with two
// cross-lane steps, then O^T = V^T P^T where MFMA i of key tile kt contracts over exactly the keys its B
// registers hold.  O goes back through the item's own q block in LDS and leaves with whole-line stores.
// reference: estimator/models/blocks/swin_layers.py:133-164 (WindowAttention.forward).
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void swin_window_attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                            const float* __restrict__ bias_table, int Hp, int Wp,
                                                                            int C, int heads, int shift) {
  constexpr int NH = 32 / HD;   // heads in a 32-channel slab
  constexpr int LD = 36;        // LDS row stride (floats)
  constexpr int ND = HD > 16 ? HD / 16 : 1;
  extern __shared__ __attribute__((aligned(16))) char swin_lds[];
  float* Qs = reinterpret_cast<float*>(swin_lds);
  float* Ks = Qs + WTOK * LD;
  float* Vs = Ks + WTOK * LD;
  int* koff = reinterpret_cast<int*>(Vs + WTOK * LD);        // per token: -4 (23 y + x), the byte step into a head's bias row
  int* region = koff + WTOK;                                 // per token: region of the shift mask
  float* bias = reinterpret_cast<float*>(region + WTOK);     // [NH][529]
  const int win = blockIdx.x, slab = blockIdx.y;
  const int nwx = Wp / WIN, nwy = Hp / WIN;
  const int wloc = win % (nwx * nwy);
  const int wy = wloc / nwx, wx = wloc % nwx;
  // only the last window row / column of a shifted layer holds more than one mask region (swin_layers.py:228-246)
  const bool masked = shift > 0 && (wy == nwy - 1 || wx == nwx - 1);
  const int t = threadIdx.x;
  const long row0 = (long)win * WTOK;
  constexpr float scale = HD == 2 ? 0.70710678118654752f : HD == 4 ? 0.5f : HD == 8 ? 0.35355339059327376f : HD == 16 ? 0.25f : 0.17677669529663688f;
  for (int idx = t; idx < WTOK * 8; idx += NW * 64) {
    const int row = idx >> 3, c4 = (idx & 7) * 4;
    const float* base = qkv + (row0 + row) * 3 * C + slab * 32 + c4;
    float4 q = *reinterpret_cast<const float4*>(base);
    const float4 k = *reinterpret_cast<const float4*>(base + C);
    const float4 v = *reinterpret_cast<const float4*>(base + 2 * C);
    q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
    *reinterpret_cast<float4*>(Qs + row * LD + c4) = q;
    *reinterpret_cast<float4*>(Ks + row * LD + c4) = k;
    *reinterpret_cast<float4*>(Vs + row * LD + c4) = v;
  }
  for (int i = t; i < NH * 529; i += NW * 64) bias[i] = bias_table[(i % 529) * heads + slab * NH + i / 529];
  if (t < WTOK) {
    const int sy = wy * WIN + t / WIN, sx = wx * WIN + t % WIN;
    const int hid = sy < Hp - WIN ? 0 : (sy < Hp - shift ? 1 : 2);
    const int wid = sx < Wp - WIN ? 0 : (sx < Wp - shift ? 1 : 2);
    koff[t] = -4 * ((t / WIN) * (2 * WIN - 1) + t % WIN);
    region[t] = shift > 0 ? hid * 3 + wid : 0;
  }
  __syncthreads();
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 15, c = lane >> 4;
  for (int item = wave; item < 9 * NH; item += NW) {
    const int qt = item % 9, hl = item / 9;
    const int ch = hl * HD;
    f32x4 acc[9];
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* Qb = Qs + (qt * 16 + r) * LD + ch + c;
    const float* Kb = Ks + r * LD + ch + c;
#pragma unroll
    for (int j = 0; j < (HD + 3) / 4; ++j) {
      const bool dok = HD >= 4 || c < HD;        // HD = 2: only k = 0, 1 carry data
      const float b = dok ? Qb[4 * j] : 0.f;
#pragma unroll
      for (int kt = 0; kt < 9; ++kt) {
        const float a = dok ? Kb[kt * 16 * LD + 4 * j] : 0.f;
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[kt], 0, 0, 0);
      }
    }
    const int tq = qt * 16 + r;
    // index(i, j) = (yi - yj + 11) * 23 + xi - xj + 11: the query's part as a byte address, the key's part from koff
    const char* bh = reinterpret_cast<const char*>(bias + hl * 529 + (WIN - 1) * (2 * WIN - 1) + WIN - 1) - koff[tq];
    const int myreg = region[tq];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
      const int4 ko = *reinterpret_cast<const int4*>(koff + kt * 16 + 4 * c);
      const int kk[4] = {ko.x, ko.y, ko.z, ko.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[kt][i] += *reinterpret_cast<const float*>(bh + kk[i]);
      if (masked) {
        const int4 rg = *reinterpret_cast<const int4*>(region + kt * 16 + 4 * c);
        const int rr[4] = {rg.x, rg.y, rg.z, rg.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rr[i] != myreg) acc[kt][i] += -100.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[kt][i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // e^(s - mx) on the hardware exp2 (1 ulp): the arguments are <= 0, the softmax sums 144 of them
    constexpr float LOG2E = 1.44269504088896340736f;
    const float mxl = mx * LOG2E;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 9; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[kt][i], LOG2E, -mxl));
        acc[kt][i] = pe;
        l += pe;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = o0;       // two chains: a dependent f32 MFMA waits 40 cycles, an independent one 32
      const bool dok = dt * 16 + r < HD;
      const float* Vb = Vs + (4 * c) * LD + ch + dt * 16 + r;
#pragma unroll
      for (int kt = 0; kt < 9; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = dok ? Vb[(kt * 16 + i) * LD] : 0.f;
          if ((kt * 4 + i) & 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[kt][i], o1, 0, 0, 0);
          else o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, acc[kt][i], o0, 0, 0, 0);
        }
      // O^T[d = 16 dt + 4 c + i][query r] -> the item's own q block (this wave is the only reader / writer of it)
      if (dt * 16 + 4 * c < HD) {
        float* dst = Qs + (qt * 16 + r) * LD + ch + dt * 16 + 4 * c;
        if constexpr (HD >= 4) {
          *reinterpret_cast<float4*>(dst) = make_float4((o0[0] + o1[0]) * inv, (o0[1] + o1[1]) * inv, (o0[2] + o1[2]) * inv, (o0[3] + o1[3]) * inv);
        } else {
          dst[0] = (o0[0] + o1[0]) * inv;
          dst[1] = (o0[1] + o1[1]) * inv;
        }
      }
    }
  }
  __syncthreads();
  for (int idx = t; idx < WTOK * 8; idx += NW * 64) {
    const int row = idx >> 3, c4 = (idx & 7) * 4;
    *reinterpret_cast<float4*>(out + (row0 + row) * C + slab * 32 + c4) = *reinterpret_cast<const float4*>(Qs + row * LD + c4);
  }
}

template <typename T>
__global__ void swin_unpartition_add_kernel(const T* __restrict__ proj, const T* __restrict__ sc, int s_ld, T* __restrict__ y,
                                            int y_ld, int B, int H, int W, int C, int shift) {
  const int Hp = (H + WIN - 1) / WIN * WIN, Wp = (W + WIN - 1) / WIN * WIN;
  const int nwx = Wp / WIN, nwy = Hp / WIN;
  const int cv = C >> 3;
  const long total = (long)B * H * W * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    long pix = i / cv;
    const int xx = (int)(pix % W);
    const int yy = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    const int sy = (yy - shift + Hp) % Hp, sx = (xx - shift + Wp) % Wp;
    const long row = (((long)b * nwy + sy / WIN) * nwx + sx / WIN) * WTOK + (sy % WIN) * WIN + (sx % WIN);
    float a[8], s8[8];
    load8(proj + row * C + v * 8, a);
    load8(sc + pix * s_ld + v * 8, s8);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += s8[e];
    store8(y + pix * y_ld + v * 8, a);
  }
}

template <typename T>
__global__ void add_rowwise_kernel(T* __restrict__ x, int x_ld, const float* __restrict__ pos, int B, int Tn, int C) {
  const int cv = C >> 3;
  const long total = (long)B * Tn * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    const long row = i / cv;
    const int t = (int)(row % Tn);
    float a[8];
    load8(x + row * x_ld + v * 8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += pos[(long)t * C + v * 8 + e];
    store8(x + row * x_ld + v * 8, a);
  }
}

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
inline int ok() { return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH; }

template <typename T>
int launch_wattn(const T* qkv, T* out, const float* bt, int B, int Hp, int Wp, int C, int heads, int shift, hipStream_t st) {
  const int hd = C / heads;
  dim3 grid(B * (Hp / WIN) * (Wp / WIN), heads);
  switch (hd) {
    case 2: hipLaunchKernelGGL((swin_window_attention_kernel<T, 2>), grid, dim3(192), 0, st, qkv, out, bt, Hp, Wp, C, heads, shift); break;
    case 4: hipLaunchKernelGGL((swin_window_attention_kernel<T, 4>), grid, dim3(192), 0, st, qkv, out, bt, Hp, Wp, C, heads, shift); break;
    case 8: hipLaunchKernelGGL((swin_window_attention_kernel<T, 8>), grid, dim3(192), 0, st, qkv, out, bt, Hp, Wp, C, heads, shift); break;
    case 16: hipLaunchKernelGGL((swin_window_attention_kernel<T, 16>), grid, dim3(192), 0, st, qkv, out, bt, Hp, Wp, C, heads, shift); break;
    case 32: hipLaunchKernelGGL((swin_window_attention_kernel<T, 32>), grid, dim3(192), 0, st, qkv, out, bt, Hp, Wp, C, heads, shift); break;
    default: return PF_ERR_ARG;
  }
  return ok();
}


// f32 rows with whole 32-channel slabs: the matrix-pipe kernel (PF_SWIN_MFMA=0 keeps the VALU kernel, for A/B and checks)
template <int HD>
int launch_wattn_mfma(const float* qkv, float* out, const float* bt, int B, int Hp, int Wp, int C, int heads, int shift, hipStream_t st) {
  constexpr int NW = 9;
  constexpr int NH = 32 / HD;
  const int lds = (3 * WTOK * 36 + NH * 529 + 2 * WTOK) * 4;
  // (the attribute belongs to a device: one bit per device and head_dim, so that a process that drives several GPUs sets it on each)
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_window_attention_mfma_kernel<HD, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return PF_ERR_LAUNCH;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  dim3 grid(B * (Hp / WIN) * (Wp / WIN), C / 32);
  hipLaunchKernelGGL((swin_window_attention_mfma_kernel<HD, NW>), grid, dim3(NW * 64), lds, st, qkv, out, bt, Hp, Wp, C, heads, shift);
  return ok();
}

inline bool swin_mfma_enabled() {          // (read per call: 24 launches per image; lets one process A/B the two kernels)
  const char* e = getenv("PF_SWIN_MFMA");
  return !(e && e[0] == '0');
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int pf_swin_ln_partition(const void* x, int x_ld, void* xw, const float* g, const float* b, float eps, int B,
                                    int H, int W, int C, int shift, int dtype, void* stream) {
  const int G = C / 8;
  if (!x || !xw || !g || !b || C % 8 || (G & (G - 1)) || G > 64 || G < 1 || x_ld % 8) return PF_ERR_ARG;
  const int Hp = (H + WIN - 1) / WIN * WIN, Wp = (W + WIN - 1) / WIN * WIN;
  const long ntok = (long)B * Hp * Wp;
  const int tpb = 256 / G;
  const int grid = grid_for(ntok, tpb);
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(swin_ln_partition_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), (const bf16_t*)x, x_ld, (bf16_t*)xw, g, b, eps, B, H, W, C, shift);
  else hipLaunchKernelGGL(swin_ln_partition_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)x, x_ld, (float*)xw, g, b, eps, B, H, W, C, shift);
  return ok();
}

extern "C" int pf_swin_window_attention(const void* qkv, void* out, const float* bias_table, int B, int Hp, int Wp, int C,
                                        int heads, int shift, int dtype, void* stream) {
  if (!qkv || !out || !bias_table || Hp % WIN || Wp % WIN || C % heads) return PF_ERR_ARG;
  if (dtype == PF_DTYPE_BF16) return launch_wattn<bf16_t>((const bf16_t*)qkv, (bf16_t*)out, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
  const int hd = C / heads;
  if (C % 32 == 0 && swin_mfma_enabled()) {
    const float* q = (const float*)qkv;
    float* o = (float*)out;
    switch (hd) {
      case 2: return launch_wattn_mfma<2>(q, o, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
      case 4: return launch_wattn_mfma<4>(q, o, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
      case 8: return launch_wattn_mfma<8>(q, o, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
      case 16: return launch_wattn_mfma<16>(q, o, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
      case 32: return launch_wattn_mfma<32>(q, o, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
      default: break;
    }
  }
  return launch_wattn<float>((const float*)qkv, (float*)out, bias_table, B, Hp, Wp, C, heads, shift, ST(stream));
}

extern "C" int pf_swin_unpartition_add(const void* proj, const void* shortcut, int s_ld, void* y, int y_ld, int B, int H,
                                       int W, int C, int shift, int dtype, void* stream) {
  if (!proj || !shortcut || !y || C % 8 || s_ld % 8 || y_ld % 8) return PF_ERR_ARG;
  const long total = (long)B * H * W * (C / 8);
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(swin_unpartition_add_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (const bf16_t*)proj, (const bf16_t*)shortcut, s_ld, (bf16_t*)y, y_ld, B, H, W, C, shift);
  else hipLaunchKernelGGL(swin_unpartition_add_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (const float*)proj, (const float*)shortcut, s_ld, (float*)y, y_ld, B, H, W, C, shift);
  return ok();
}

extern "C" int pf_add_rowwise(void* x, int x_ld, const float* pos, int B, int T, int C, int dtype, void* stream) {
  if (!x || !pos || C % 8 || x_ld % 8) return PF_ERR_ARG;
  const long total = (long)B * T * (C / 8);
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(add_rowwise_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (bf16_t*)x, x_ld, pos, B, T, C);
  else hipLaunchKernelGGL(add_rowwise_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (float*)x, x_ld, pos, B, T, C);
  return ok();
}
