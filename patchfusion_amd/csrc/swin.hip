// G2L (Swin) kernels: LayerNorm + pad + cyclic shift + window partition, the 144-token window
// attention with relative-position bias and shift mask, and window reverse + residual.
// head_dim is tiny here (2..32) -- MFMA-hostile -- so the attention runs on the VALU with K/V of one
// (window, head) in LDS (broadcast reads) and one query per thread; it is HBM/LDS bound.
#include "pf_common.h"
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
