// Common device helpers for the PatchFusion gfx950 kernels.
// Activation element type T is either float ("exact" mode, f32-input MFMA) or bf16 stored as raw
// uint16 ("fast" mode, bf16 MFMA with f32 accumulation).  All tensors are NHWC ("pixel-major"):
// a pixel's channels are contiguous, `ld` = elements between consecutive pixels (>= channels,
// multiple of 8) so producers can write straight into channel slices of concat buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint16_t bf16_t;

#define PF_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (inputs are finite)
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16x2 (round to nearest even) in ONE instruction; gfx950 has no clang builtin for it
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 8 consecutive channels as floats (two 16-B loads for float, one for bf16)
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  uint4 a = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 a;
  a.x = cvt_pk_bf16(v[0], v[1]);
  a.y = cvt_pk_bf16(v[2], v[3]);
  a.z = cvt_pk_bf16(v[4], v[5]);
  a.w = cvt_pk_bf16(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = a;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
// (plain C++ packing here: the MFMA epilogues use store4, and an inline-asm pack makes hipcc spill the
// accumulators of the register-tight 3x3 halo kernels - 832 B/lane of scratch, 2.5x slower end to end)
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  uint2 r;
  r.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  r.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  *reinterpret_cast<uint2*>(p) = r;
}

// split-precision ("f32x3") producers, csrc/gemm_split3.hip: x = h + m + l in three bf16 planes.
// Two floats -> their three packed bf16x2 words, on v_cvt_pk_bf16_f32 (round to nearest even = f2bf for finite values): 11 instructions per pair
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
#pragma clang fp contract(off)      // a - h must subtract from the ROUNDED value a, not fuse with the multiply that produced it
  h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(ra, rb);
  l = cvt_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void store_split3(bf16_t* y, long plane, const float (&v)[4]) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split3_pair(v[0], v[1], h0, m0, l0);
  split3_pair(v[2], v[3], h1, m1, l1);
  *reinterpret_cast<uint2*>(y) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(y + plane) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(y + 2 * plane) = make_uint2(l0, l1);
}

// element offset of (row, col) in one plane of a split-precision operand: row-major [rows][ld] (kmaj_rows == 0) or CHUNK-MAJOR
// [cols/32][kmaj_rows][32] (csrc/gemm_split3.hip: every 32-deep K chunk of all rows is one contiguous slab); col % 4 == 0 runs stay inside a chunk
__device__ __forceinline__ long split3_at(long row, int col, int ld, long kmaj_rows) {
  return kmaj_rows ? ((long)(col >> 5) * kmaj_rows + row) * 32 + (col & 31) : row * ld + col;
}

__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  uint2 a = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}

// exact-erf GELU (torch.nn.GELU()).  erf(a) = 1 - 2^(-a P(a)) on a = min(|x| / sqrt 2, 4) with a degree-8 P fitted to -log2(erfc(a)) / a
// (round 6; tools/fit_erf.py): one branch-free chain of eight FMAs and one hardware exp2 where the library's erff evaluates two polynomial
// branches under divergence.  max |gelu - float64 gelu| over [-6, 6] = 4.5e-7, the same as with a correctly rounded float32 erf (the
// error is the rounding of 1 + erf).  -DPF_GELU_OCML keeps the library call (A/B builds).
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef PF_GELU_OCML
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
  const float a = fminf(fabsf(x) * 0.70710678118654752440f, 4.0f);
  float p = -2.8146364456915762e-06f;
  p = __builtin_fmaf(p, a, 5.093829895486124e-05f);
  p = __builtin_fmaf(p, a, -0.00036259577609598637f);
  p = __builtin_fmaf(p, a, 0.0010585421696305275f);
  p = __builtin_fmaf(p, a, 0.0016202160622924566f);
  p = __builtin_fmaf(p, a, -0.029034368693828583f);
  p = __builtin_fmaf(p, a, 0.14880506694316864f);
  p = __builtin_fmaf(p, a, 0.9183712005615234f);
  p = __builtin_fmaf(p, a, 1.6279088258743286f);
  const float e = 1.0f - __builtin_amdgcn_exp2f(-(p * a));
  return 0.5f * x * (1.0f + copysignf(e, x));
#endif
}
// torch.nn.Softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a linear block id: consecutive logical ids land on one XCD
// (block b is observed to run on XCD b % 8; used for L2 locality only, never for correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int NX = 8;
  int q = nblocks / NX, r = nblocks % NX;
  int xcd = bid % NX, idx = bid / NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// status codes of the C ABI
#define PF_OK 0
#define PF_ERR_ARG 1
#define PF_ERR_LAUNCH 2
