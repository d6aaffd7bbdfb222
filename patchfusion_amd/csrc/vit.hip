// ViT (DINOv2) encoder kernels other than the linear layers (those run on igemm.hip):
// patch im2col + ImageNet normalisation, token assembly, LayerNorm, qkv head split and the fused
// softmax attention on the matrix cores.
#include <type_traits>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// im2col for the 14x14/14 patch embedding, fused with (x - mean) / std.
// out[(b,ty,tx)][(ky*14+kx)*3 + c], columns 588..ld-1 zero.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void patch_im2col_kernel(const float* __restrict__ img, int B, int H, int W, T* __restrict__ out, int ld) {
  const int th = H / 14, tw = W / 14;
  const long total = (long)B * th * tw * ld;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % ld);
    const long row = i / ld;
    float v = 0.f;
    if (k < 588) {
      const int c = k % 3, kk = k / 3, kx = kk % 14, ky = kk / 14;
      const int tx = (int)(row % tw), ty = (int)((row / tw) % th), b = (int)(row / ((long)tw * th));
      const float px = img[(((long)b * 3 + c) * H + (ty * 14 + ky)) * W + (tx * 14 + kx)];
      v = (px - mean[c]) / stdv[c];
    }
    Elem<T>::st(out + i, v);
  }
}

// tokens[b,0,:] = cls + pos[0]; tokens[b,1+t,:] = emb[b*(S-1)+t,:] + pos[1+t]
template <typename T>
__global__ void assemble_tokens_kernel(const T* __restrict__ emb, T* __restrict__ tok, const float* __restrict__ cls,
                                       const float* __restrict__ pos, int B, int S, int D) {
  const long total = (long)B * S * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int s = (int)((i / D) % S);
    const int b = (int)(i / ((long)D * S));
    float v = (s == 0) ? cls[d] : Elem<T>::ld(emb + ((long)b * (S - 1) + (s - 1)) * D + d);
    Elem<T>::st(tok + i, v + pos[(long)s * D + d]);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, D <= 2048, two-pass statistics in f32.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        float eps, int batches, int in_rpb, int in_off, int out_rpb, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)batches * out_rpb) return;
  const int b = (int)(row / out_rpb), t = (int)(row % out_rpb);
  const T* xr = x + ((long)b * in_rpb + in_off + t) * x_ld;
  T* yr = y + row * y_ld;
  const int nv = D >> 3;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nv) {
      load8(xr + vi * 8, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nv) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[vi * 8 + e] + bta[vi * 8 + e];
      store8(yr + vi * 8, o);
    }
  }
}

// LayerNorm of float32 rows with the output as three bf16 planes (x = h + m + l) for the split-precision GEMM (csrc/gemm_split3.hip)
// kmaj_rows != 0: the planes are chunk-major [D/32][kmaj_rows][32] (pf_common.h split3_at)
__global__ __launch_bounds__(256) void layernorm_split3_kernel(const float* __restrict__ x, int x_ld, bf16_t* __restrict__ y, int y_ld, long plane,
                                                               const float* __restrict__ g, const float* __restrict__ bta, float eps, long rows, int D,
                                                               long kmaj_rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * x_ld;
  bf16_t* yr = y + row * y_ld;
  const int nv = D >> 3;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nv) {
      load8(xr + vi * 8, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = lane + 64 * i;
    if (vi < nv) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[vi * 8 + e] + bta[vi * 8 + e];
      const float lo[4] = {o[0], o[1], o[2], o[3]}, hi[4] = {o[4], o[5], o[6], o[7]};
      bf16_t* yo = kmaj_rows ? y + split3_at(row, vi * 8, y_ld, kmaj_rows) : yr + vi * 8;
      store_split3(yo, plane, lo);
      store_split3(yo + 4, plane, hi);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// qkv split: qkv [B*S][3][Hh][64] -> Q*scale [B,Hh,S,64], K [B,Hh,S,64], V^T [B,Hh,64,Sp]
// grid (ceil(S/64), B*Hh), 256 threads.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void qkv_split_kernel(const T* __restrict__ qkv, int B, int S, int Hh, T* __restrict__ q,
                                                        T* __restrict__ k, T* __restrict__ vt, int Sp, float scale) {
  __shared__ float vs[64][65];
  const int D = Hh * 64;
  const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
  const int s0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  // 64 tokens x 8 vectors-of-8 per matrix
  for (int i = tid; i < 64 * 8; i += 256) {
    const int t = i >> 3, vj = i & 7;
    const int s = s0 + t;
    float a[8];
    if (s < S) {
      const T* src = qkv + ((long)b * S + s) * 3 * D + h * 64 + vj * 8;
      load8(src, a);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] *= scale;
      store8(q + (((long)bh * S + s) * 64) + vj * 8, a);
      load8(src + D, a);
      store8(k + (((long)bh * S + s) * 64) + vj * 8, a);
      load8(src + 2 * D, a);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) vs[t][vj * 8 + e] = a[e];
  }
  __syncthreads();
  {
    const int d = tid >> 2, tc = (tid & 3) * 16;
    T* dst = vt + ((long)bh * 64 + d) * Sp + s0 + tc;
    float a[8];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = vs[tc + hlf * 8 + e][d];
      store8(dst + hlf * 8, a);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused attention, head_dim 64.  Block = 4 waves, each wave 16 queries; KV tiles of 64 keys staged
// in LDS (K as [key][d], V^T as [d][key], 16-byte slots XOR-swizzled).  Scores are computed
// TRANSPOSED (S^T = K Q^T) so that a lane's 16 score registers all belong to ONE query: the row
// max / sum are in-lane plus two cross-lane-group shuffles, and exp(S^T) is directly the MFMA "B"
// operand of O^T = V^T P^T.
// ---------------------------------------------------------------------------------------------
template <typename T> struct AttnCfg;
template <> struct AttnCfg<bf16_t> { static constexpr int ROWB = 128, SPR = 8; };
template <> struct AttnCfg<float> { static constexpr int ROWB = 256, SPR = 16; };

template <typename T> __device__ __forceinline__ int attn_swz(int row, int slot) {
  if constexpr (sizeof(T) == 2) return slot ^ ((row >> 1) & 7);
  else return slot ^ (row & 15);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void vit_attention_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ vt, T* __restrict__ out, int B, int S,
                                                            int Sp, int Hh) {
  constexpr int ROWB = AttnCfg<T>::ROWB, SPR = AttnCfg<T>::SPR;
  constexpr int VEC = Elem<T>::VEC;
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * ROWB];
  char* Ks = lds;
  char* Vs = lds + 64 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int D = Hh * 64;
  const T* qb = q + (long)bh * S * 64;
  const T* kb = k + (long)bh * S * 64;
  const T* vb = vt + (long)bh * 64 * Sp;

  // Q fragments for query q0 + r (clamped; out-of-range queries are never stored)
  const int qi = min(q0 + r, S - 1);
  uint4 qf[sizeof(T) == 2 ? 2 : 4];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qb + (long)qi * 64 + ks * 32 + g * 8);
  } else {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *reinterpret_cast<const uint4*>(qb + (long)qi * 64 + s4 * 16 + g * 4);
  }

  f32x4 o[4];
#pragma unroll
  for (int fd = 0; fd < 4; ++fd) o[fd] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (S + 63) / 64;
  // K tile [64 keys][64 d] and V^T tile [64 d][64 keys] go global -> registers -> LDS; the global loads of tile kt+1 are issued
  // BEFORE tile kt is multiplied, so their latency overlaps the 128 MFMAs of the tile instead of sitting between two barriers
  constexpr int RPP = 256 / SPR, NPASS = 64 / RPP;        // rows per pass, passes
  const int sj = tid % SPR, srr0 = tid / SPR;
  typedef unsigned __attribute__((ext_vector_type(4))) u32x4v;   // (native vector: HIP's uint4 struct arrays ended up in scratch here)
  u32x4v kreg[NPASS], vreg[NPASS];
#define PF_ATTN_GLOAD(KT)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < NPASS; ++i) {                                                       \
    const int row = srr0 + RPP * i;                                                                         \
    const int key = (KT) * 64 + row;                                                                        \
    const u32x4v kv = *reinterpret_cast<const u32x4v*>(kb + (long)min(key, S - 1) * 64 + sj * VEC);        \
    kreg[i] = key < S ? kv : u32x4v{0u, 0u, 0u, 0u};                                                        \
    vreg[i] = *reinterpret_cast<const u32x4v*>(vb + (long)row * Sp + (KT) * 64 + sj * VEC);                 \
  }
  PF_ATTN_GLOAD(0)
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();                                      // every wave has finished reading tile kt-1
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = srr0 + RPP * i;
      *reinterpret_cast<u32x4v*>(Ks + row * ROWB + (attn_swz<T>(row, sj) << 4)) = kreg[i];
      *reinterpret_cast<u32x4v*>(Vs + row * ROWB + (attn_swz<T>(row, sj) << 4)) = vreg[i];
    }
    __syncthreads();
    if (kt + 1 < ntiles) { PF_ATTN_GLOAD(kt + 1) }
    // ---- S^T = K Q^T ----
    f32x4 sc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) sc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int row = f * 16 + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint4 a = *reinterpret_cast<const uint4*>(Ks + row * ROWB + (attn_swz<T>(row, ks * 4 + g) << 4));
          sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[ks]),
                                                         sc[f], 0, 0, 0);
        }
      }
    } else {
      // the four key fragments are four INDEPENDENT accumulator chains: walk them round-robin (f innermost) so that no
      // v_mfma_f32_16x16x4_f32 waits for its predecessor (40-cycle dependent latency against a 32-cycle issue interval)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        uint4 a[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) a[f] = *reinterpret_cast<const uint4*>(Ks + (f * 16 + r) * ROWB + (attn_swz<T>(f * 16 + r, s4 * 4 + g) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const uint32_t av = e == 0 ? a[f].x : e == 1 ? a[f].y : e == 2 ? a[f].z : a[f].w;
            const uint32_t qv = e == 0 ? qf[s4].x : e == 1 ? qf[s4].y : e == 2 ? qf[s4].z : qf[s4].w;
            sc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av), __uint_as_float(qv), sc[f], 0, 0, 0);
          }
      }
    }
    // ---- online softmax for query (lane & 15); this lane holds keys kt*64 + f*16 + g*4 + e ----
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = kt * 64 + f * 16 + g * 4 + e;
        if (key >= S) sc[f][e] = -INFINITY;
        mx = fmaxf(mx, sc[f][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = expf(sc[f][e] - m_new);
        sc[f][e] = pe;
        psum += pe;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int fd = 0; fd < 4; ++fd)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[fd][e] *= alpha;
    // ---- O^T += V^T P^T ----
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        uint4 pb;
        pb.x = (uint32_t)f2bf(sc[2 * jj][0]) | ((uint32_t)f2bf(sc[2 * jj][1]) << 16);
        pb.y = (uint32_t)f2bf(sc[2 * jj][2]) | ((uint32_t)f2bf(sc[2 * jj][3]) << 16);
        pb.z = (uint32_t)f2bf(sc[2 * jj + 1][0]) | ((uint32_t)f2bf(sc[2 * jj + 1][1]) << 16);
        pb.w = (uint32_t)f2bf(sc[2 * jj + 1][2]) | ((uint32_t)f2bf(sc[2 * jj + 1][3]) << 16);
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
          const int row = fd * 16 + r;
          // keys (2jj)*16 + g*4 .. +3 -> byte (2jj)*32 + g*8 ; keys (2jj+1)*16 + g*4 .. -> byte (2jj+1)*32 + g*8
          const int slot0 = (2 * jj) * 2 + (g >> 1), slot1 = (2 * jj + 1) * 2 + (g >> 1);
          const uint2 lo = *reinterpret_cast<const uint2*>(Vs + row * ROWB + (attn_swz<T>(row, slot0) << 4) + (g & 1) * 8);
          const uint2 hi = *reinterpret_cast<const uint2*>(Vs + row * ROWB + (attn_swz<T>(row, slot1) << 4) + (g & 1) * 8);
          const uint4 a = make_uint4(lo.x, lo.y, hi.x, hi.y);
          o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, pb), o[fd],
                                                         0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        uint4 a[4];
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) a[fd] = *reinterpret_cast<const uint4*>(Vs + (fd * 16 + r) * ROWB + (attn_swz<T>(fd * 16 + r, f * 4 + g) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int fd = 0; fd < 4; ++fd) {      // four independent accumulator chains, round-robin (see the QK loop)
            const uint32_t av = e == 0 ? a[fd].x : e == 1 ? a[fd].y : e == 2 ? a[fd].z : a[fd].w;
            o[fd] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av), sc[f][e], o[fd], 0, 0, 0);
          }
      }
    }
  }
#undef PF_ATTN_GLOAD
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  const int qo = q0 + r;
  if (qo < S) {
    T* dst = out + ((long)b * S + qo) * D + h * 64 + g * 4;
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) store4(dst + fd * 16, o[fd][0] * inv, o[fd][1] * inv, o[fd][2] * inv, o[fd][3] * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 attention, 32 queries per wave on v_mfma_f32_32x32x16_bf16 (the f32 path keeps the 16-query kernel
// above).  Block = 4 waves = 128 queries; KV tiles of 64 keys, double-buffered in LDS (one barrier per tile,
// next tile's global loads in flight during the MFMAs).  S^T = K Q^T: lane (q = l&31, h = l>>5) holds for its
// query the keys f*32 + 8j + 4h + i (register 4j+i of fragment f), so max/sum are in-lane + one shuffle, and
// registers 8jj..8jj+7 of fragment f are exactly the 8 bf16 the PV MFMA wants as its B operand for the
// 16-key step (f,jj) - with the V^T fragment reading the same keys (two 8-byte reads).
// LDS: K rows 128 B, slot ^= (row>>1)&7 (conflict-free b128 fragment reads); V^T rows 128 B with the same
// slot swizzle plus the two 8-byte halves of a slot swapped for rows 16..31 (mod 32), which makes the 8-byte
// fragment reads of a 32-lane group hit 32 distinct 8-byte bank pairs.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ __launch_bounds__(256) void vit_attention32_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                              const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int B,
                                                              int S, int Sp, int Hh) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * 64 * 128];   // [stage][K | V^T][64 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int D = Hh * 64;
  const bf16_t* qb = q + (long)bh * S * 64;
  const bf16_t* kb = k + (long)bh * S * 64;
  const bf16_t* vb = vt + (long)bh * 64 * Sp;

  const int qi = min(q0 + fr, S - 1);
  uint4 qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const uint4*>(qb + (long)qi * 64 + t * 16 + fh * 8);

  f32x16_t o[2];
#pragma unroll
  for (int fd = 0; fd < 2; ++fd)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[fd][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: 64 rows x 8 vectors per matrix = 512 vectors; thread -> rows (tid>>3) and (tid>>3)+32, vector tid&7
  const int sj = tid & 7, sr = tid >> 3;
  uint4 kreg[2], vreg[2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = sr + 32 * i;
      const int key = kt * 64 + row;
      kreg[i] = make_uint4(0, 0, 0, 0);
      if (key < S) kreg[i] = *reinterpret_cast<const uint4*>(kb + (long)key * 64 + sj * 8);
      vreg[i] = *reinterpret_cast<const uint4*>(vb + (long)row * Sp + kt * 64 + sj * 8);
    }
  };
  auto lstore = [&](int stage) {
    char* Ks = lds + stage * (2 * 64 * 128);
    char* Vs = Ks + 64 * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = sr + 32 * i;
      const int ps = (sj ^ ((row >> 1) & 7)) << 4;
      *reinterpret_cast<uint4*>(Ks + row * 128 + ps) = kreg[i];
      uint4 v = vreg[i];
      if ((row >> 4) & 1) v = make_uint4(v.z, v.w, v.x, v.y);   // swap the 8-byte halves
      *reinterpret_cast<uint4*>(Vs + row * 128 + ps) = v;
    }
  };

  const int ntiles = (S + 63) / 64;
  gload(0);
  lstore(0);
  __syncthreads();
  const int kswz = (fr >> 1) & 7;          // fragment rows are f*32 + fr
  const int vhalf = ((fr >> 4) & 1);
  for (int kt = 0; kt < ntiles; ++kt) {
    const bool more = kt + 1 < ntiles;
    if (more) gload(kt + 1);
    const char* Ks = lds + (kt & 1) * (2 * 64 * 128);
    const char* Vs = Ks + 64 * 128;
    // ---- S^T = K Q^T : 2 key fragments x 4 d-steps ----
    f32x16_t sc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[f][e] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint4 a = *reinterpret_cast<const uint4*>(Ks + (f * 32 + fr) * 128 + ((((t << 1) | fh) ^ kswz) << 4));
        sc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[t]), sc[f], 0, 0, 0);
      }
    }
    // ---- online softmax (query = lane & 31); register 4j+i of fragment f is key kt*64 + f*32 + 8j + 4fh + i ----
    if (kt == ntiles - 1) {   // only the last tile can contain keys >= S (wave-uniform branch)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 64 + f * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
          if (key >= S) sc[f][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // statistics in the base-2 domain: exp(x - m) = exp2(x*log2e - m*log2e); one fma + one v_exp_f32 per score
    constexpr float LOG2E = 1.4426950408889634f;
    const float m_new = fmaxf(m_run, mx * LOG2E);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f(fmaf(sc[f][r], LOG2E, -m_new));
        sc[f][r] = pe;
        psum += pe;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int fd = 0; fd < 2; ++fd)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[fd][e] *= alpha;
    // ---- O^T += V^T P^T : 4 steps of 16 keys (f, jj) x 2 d-fragments ----
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        uint4 pb;
        pb.x = cvt_pk_bf16(sc[f][8 * jj + 0], sc[f][8 * jj + 1]);
        pb.y = cvt_pk_bf16(sc[f][8 * jj + 2], sc[f][8 * jj + 3]);
        pb.z = cvt_pk_bf16(sc[f][8 * jj + 4], sc[f][8 * jj + 5]);
        pb.w = cvt_pk_bf16(sc[f][8 * jj + 6], sc[f][8 * jj + 7]);
        // keys f*32+16jj+4fh..+3 -> logical slot 4f+2jj, 8-byte half fh ; keys +8 -> slot 4f+2jj+1, half fh
        const int s0 = 4 * f + 2 * jj;
#pragma unroll
        for (int fd = 0; fd < 2; ++fd) {
          const char* rowp = Vs + (fd * 32 + fr) * 128 + ((fh ^ vhalf) << 3);
          const uint2 lo = *reinterpret_cast<const uint2*>(rowp + ((s0 ^ kswz) << 4));
          const uint2 hi = *reinterpret_cast<const uint2*>(rowp + (((s0 + 1) ^ kswz) << 4));
          const uint4 a = make_uint4(lo.x, lo.y, hi.x, hi.y);
          o[fd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, pb), o[fd], 0, 0, 0);
        }
      }
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }
  float l = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l;
  const int qo = q0 + fr;
  if (qo < S) {
    bf16_t* dst = out + ((long)b * S + qo) * D + h * 64 + 4 * fh;
#pragma unroll
    for (int fd = 0; fd < 2; ++fd)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        store4(dst + fd * 32 + 8 * j, o[fd][4 * j] * inv, o[fd][4 * j + 1] * inv, o[fd][4 * j + 2] * inv, o[fd][4 * j + 3] * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// f32 attention, version 2 (round 3): reads Q, K, V STRAIGHT from the QKV GEMM's output rows [B*S][3][Hh][64] -- no qkv_split
// launch, no Q / K / V^T buffers -- and spends far fewer VALU instructions per key tile.  With v_mfma_f32_16x16x4_f32 the VALU work
// of a SIMD does not overlap its matrix pipe (they serialise: round-3 s_memtime timelines of csrc/wino_fused.hip), so the 526 VALU
// instructions per 64-key tile of vit_attention_kernel<float> (full-precision expf, per-score key masking) cost as much pipe time as
// half of the tile's 128 MFMAs.  Here: scores in the base-2 domain (Q is scaled by head_dim^-1/2 * log2 e when its fragments are
// loaded; one v_exp_f32 per score), key masking only in the last tile (wave-uniform branch), V transposed on its way into LDS
// (global [key][d] -> LDS [d][key], 4-byte stores) so that the P.V fragment reads stay 16-byte.
// Same structure otherwise: block = 4 waves x 16 queries, S^T = K Q^T so that a lane's 16 scores belong to ONE query, online softmax
// with f32 statistics, O^T = V^T P^T.
// ---------------------------------------------------------------------------------------------
template <int QG>      // query groups of 16 per wave: a block covers 64 QG queries; the K / V^T fragments are read once for all groups
__global__ __launch_bounds__(256, QG == 1 ? 3 : 2) void vit_attention_qkv_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int S,
                                                                     int Hh, float qscale, long plane, long kmaj_rows) {
  constexpr int ROWB = 256;
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * ROWB];
  char* Ks = lds;                       // [key][d]  16-byte slot ^ (key & 15)
  char* Vs = lds + 64 * ROWB;           // [d][key]  16-byte slot ^ vswz(d)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
  const int q0 = blockIdx.x * (64 * QG) + wave * (16 * QG);
  const bool has_q = q0 < S;                                       // (wave-uniform) the last query block of S = 1037 keeps one wave of four
  const int D = Hh * 64;
  const long rs = 3L * D;                                          // row stride of the qkv matrix
  const float* base = qkv + (long)b * S * rs + h * 64;             // + s * rs (+ D for K, + 2 D for V)

  float4 qf[QG][4];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const int qi = min(q0 + qg * 16 + r, S - 1);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      qf[qg][s4] = *reinterpret_cast<const float4*>(base + (long)qi * rs + s4 * 16 + g * 4);
      qf[qg][s4].x *= qscale; qf[qg][s4].y *= qscale; qf[qg][s4].z *= qscale; qf[qg][s4].w *= qscale;
    }
  }
  f32x4 o[QG][4];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    m_run[qg] = -INFINITY;
    l_run[qg] = 0.f;
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) o[qg][fd] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int ntiles = (S + 63) / 64;
  const int nf_last = (S - (ntiles - 1) * 64 + 15) >> 4;           // 16-key fragments of the last tile that hold keys (1037: one of four)
  // staging: 64 keys x 16 vectors of 4 floats per matrix; thread -> vector sj = tid % 16 of keys srr0 + 16 i.
  // V^T swizzle vswz(d) = (d & 15) ^ ((d >> 4) & 3): a bijection of d & 15 for the 16-row fragment reads (conflict free) that also
  // spreads the transposing 4-byte stores (d = 4 sj + e over the 16 lanes of a key) over 8 slots instead of 2 (PMC of the first
  // version: 53 % of the LDS cycles were bank conflicts)
  const int sj = tid & 15, srr0 = tid >> 4;
  float4 kreg[4], vreg[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = kt * 64 + srr0 + 16 * i;
      const float* rp = base + (long)min(key, S - 1) * rs + sj * 4;
      const float4 kv = *reinterpret_cast<const float4*>(rp + D);
      const float4 vv = *reinterpret_cast<const float4*>(rp + 2 * D);
      const bool ok = key < S;
      kreg[i] = ok ? kv : make_float4(0.f, 0.f, 0.f, 0.f);
      vreg[i] = ok ? vv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // one key tile for this wave's 16 QG queries: FULL (no masking, four fragments) or the last tile (nfr fragments, masked)
  auto tile = [&](int kt, auto full_c, int nfr) {
    constexpr bool FULL = decltype(full_c)::value;
    // ---- S^T = K Q^T (base-2 logits): 4 QG independent accumulator chains, round-robin ----
    f32x4 sc[QG][4];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
      for (int f = 0; f < 4; ++f) sc[qg][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float4 a[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) a[f] = *reinterpret_cast<const float4*>(Ks + (f * 16 + r) * ROWB + (((s4 * 4 + g) ^ r) << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (!FULL && f >= nfr) continue;
          const float av = e == 0 ? a[f].x : e == 1 ? a[f].y : e == 2 ? a[f].z : a[f].w;
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) {
            const float qv = e == 0 ? qf[qg][s4].x : e == 1 ? qf[qg][s4].y : e == 2 ? qf[qg][s4].z : qf[qg][s4].w;
            sc[qg][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, qv, sc[qg][f], 0, 0, 0);
          }
        }
    }
    // ---- online softmax for query (group qg, lane & 15); this lane holds keys kt*64 + f*16 + g*4 + e ----
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      if (!FULL) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (kt * 64 + f * 16 + g * 4 + e >= S) sc[qg][f][e] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(fmaxf(sc[qg][0][0], sc[qg][0][1]), fmaxf(sc[qg][0][2], sc[qg][0][3])),
                       fmaxf(fmaxf(sc[qg][1][0], sc[qg][1][1]), fmaxf(sc[qg][1][2], sc[qg][1][3])));
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sc[qg][2][0], sc[qg][2][1]), fmaxf(sc[qg][2][2], sc[qg][2][3])),
                           fmaxf(fmaxf(sc[qg][3][0], sc[qg][3][1]), fmaxf(sc[qg][3][2], sc[qg][3][3]))));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qg], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qg] - m_new);
      m_run[qg] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = __builtin_amdgcn_exp2f(sc[qg][f][e] - m_new);
          sc[qg][f][e] = pe;
          psum += pe;
        }
      l_run[qg] = l_run[qg] * alpha + psum;
#pragma unroll
      for (int fd = 0; fd < 4; ++fd)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[qg][fd][e] *= alpha;
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (!FULL && f >= nfr) continue;
      float4 a[4];
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) a[fd] = *reinterpret_cast<const float4*>(Vs + (fd * 16 + r) * ROWB + (((f * 4 + g) ^ r ^ fd) << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
          const float av = e == 0 ? a[fd].x : e == 1 ? a[fd].y : e == 2 ? a[fd].z : a[fd].w;
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) o[qg][fd] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sc[qg][f][e], o[qg][fd], 0, 0, 0);
        }
    }
  };
  gload(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();                                      // every wave has finished reading tile kt-1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = srr0 + 16 * i;                      // row of K, column of V^T
      *reinterpret_cast<float4*>(Ks + key * ROWB + ((sj ^ (key & 15)) << 4)) = kreg[i];
      const float ve[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = sj * 4 + e;
        *reinterpret_cast<float*>(Vs + d * ROWB + (((key >> 2) ^ (d & 15) ^ ((d >> 4) & 3)) << 4) + (key & 3) * 4) = ve[e];
      }
    }
    __syncthreads();
    if (kt + 1 < ntiles) gload(kt + 1);
    if (has_q) {
      if (kt + 1 < ntiles) tile(kt, std::true_type{}, 4);
      else tile(kt, std::false_type{}, nf_last);
    }
  }
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    float l = l_run[qg];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qo = q0 + qg * 16 + r;
    if (qo < S) {
      const long at = ((long)b * S + qo) * D + h * 64 + g * 4;
      if (plane) {                  // out = three bf16 planes for the split-precision projection GEMM (csrc/gemm_split3.hip)
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
          const float w4[4] = {o[qg][fd][0] * inv, o[qg][fd][1] * inv, o[qg][fd][2] * inv, o[qg][fd][3] * inv};
          store_split3(reinterpret_cast<bf16_t*>(out) + split3_at((long)b * S + qo, h * 64 + g * 4 + fd * 16, D, kmaj_rows), plane, w4);
        }
      } else {
        float* dst = out + at;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) store4(dst + fd * 16, o[qg][fd][0] * inv, o[qg][fd][1] * inv, o[qg][fd][2] * inv, o[qg][fd][3] * inv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// float32-grade ViT attention on the bf16 matrix cores (split precision, csrc/gemm_split3.hip): q, k, v arrive as THREE bf16 planes each
// (the QKV GEMM writes them: x = h + m + l exactly), S^T = K.Q^T and O^T = V^T.P^T are evaluated as the six leading partial products on
// v_mfma_f32_16x16x32_bf16 with float32 accumulation, and the softmax probabilities are split in registers.  Against the f32-MFMA kernel
// above the matrix work is 2.67x faster AND no longer shares its issue port with the softmax VALU work (the f32 MFMA does).
//
// Block = 4 waves x 16 queries, KV tiles of 64 keys, single LDS stage of 48 KB (three blocks per CU):
//   Ks[p][key][64 d]   bf16, 128-byte rows, 16-byte slot s of row key at slot s ^ (key & 7)
//   Vs[p][d][64 keys]  bf16, transposed while staging; inside each 32-key block key 16 a + 4 g + t sits at position 8 g + 4 a + t, so that
//                      the eight keys a lane holds of P (two accumulator quads: keys 4 g + t of two 16-key fragments) are ONE 16-byte read
//                      of V^T; 16-byte slot s of row d at slot s ^ (d & 7) ^ (((d >> 4) & 3) << 1).  (The second term is round 4: the transposing
//                      4-byte stores of a 32-lane group go to rows d = 8 slot + e, slot = 0..7 -- without it all eight hit ONE bank (PMC: 54 % of the
//                      kernel's LDS cycles were bank conflicts); with it four banks, two lanes each, which a ds_write_b32 absorbs.  For the fragment
//                      reads, d = 16 fd + r, it is a per-instruction constant: they stay conflict-free.)
// S^T fragment kf (16 keys): A = K rows (lane (r = key, g): d = 32 kh + 8 g ..), B = Q (lane (r = query, g): same d) -> lane (query r, g)
// holds keys 16 kf + 4 g + e.  O^T fragment df (16 d): A = V^T rows (lane (r = d, g): the eight permuted keys of block kk), B = P.
// ---------------------------------------------------------------------------------------------------------------------------------------
// (Measured and NOT kept, round 5: loading the K / V rows of key tile kt+1 into the 48 staging registers BEFORE the MFMA / softmax work of tile kt --
// 217 registers, two blocks per CU instead of three: 310 vs 285 us at B = 8, the image pass +1.2 ms, profiles/r5_attention_prefetch.md.  Three
// resident blocks cover the load latency better than a prefetch in two.)
__global__ __launch_bounds__(256, 3) void vit_attention_split3_kernel(const bf16_t* __restrict__ qkv3, long plane_in, bf16_t* __restrict__ out3,
                                                                      long plane_out, int B, int S, int Hh, float qscale, long kmaj_rows) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 3 * 64 * 128];
  char* Ks = lds;
  char* Vs = lds + 3 * 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool has_q = q0 < S;
  const int D = Hh * 64;
  const long rs = 3L * D;
  const bf16_t* base = qkv3 + (long)b * S * rs + h * 64;            // + plane * plane_in + s * rs (+ D for K, + 2 D for V)

  uint4 qf[3][2];
  {
    const int qi = min(q0 + r, S - 1);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) qf[pl][kh] = *reinterpret_cast<const uint4*>(base + pl * plane_in + (long)qi * rs + 32 * kh + 8 * g);
  }
  f32x4 o[4];
#pragma unroll
  for (int fd = 0; fd < 4; ++fd) o[fd] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (S + 63) / 64;

  // staging roles.  K: item (plane, key, slot): 3 x 64 x 8 = 1536 sixteen-byte items, 6 per thread.  V: item (plane, key pair, slot of 8 d):
  // 3 x 32 x 8 = 768, 3 per thread: two 16-byte loads (keys 2 j, 2 j + 1), eight 4-byte transposing stores (positions of a key pair are adjacent)
  uint4 kreg[6], vreg[3][2];
  // per-thread source rows as 32-bit element offsets from one wave-uniform base that advances by a tile per iteration (64-bit pointers per
  // row cost 9 more registers: the kernel sits at the 168-register limit of three blocks per CU and used to spill); masks only in the last tile
  unsigned koff[6], voff[3];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int it = tid + 256 * i;
    koff[i] = (unsigned)((it >> 9) * plane_in + (long)((it >> 3) & 63) * rs + D + 8 * (it & 7));
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int it = tid + 256 * i;
    voff[i] = (unsigned)((it >> 8) * plane_in + (long)(2 * ((it >> 3) & 31)) * rs + 2 * D + 8 * (it & 7));
  }
  const bf16_t* tbase = base;
  const long tile_step = 64 * rs;
  auto gload = [&](int kt) {
    if (kt + 1 < ntiles) {
#pragma unroll
      for (int i = 0; i < 6; ++i) kreg[i] = *reinterpret_cast<const uint4*>(tbase + koff[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        vreg[i][0] = *reinterpret_cast<const uint4*>(tbase + voff[i]);
        vreg[i][1] = *reinterpret_cast<const uint4*>(tbase + voff[i] + rs);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int kg = kt * 64 + (((tid + 256 * i) >> 3) & 63);
        kreg[i] = make_uint4(0, 0, 0, 0);
        if (kg < S) kreg[i] = *reinterpret_cast<const uint4*>(tbase + koff[i]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int kg = kt * 64 + 2 * (((tid + 256 * i) >> 3) & 31);
        vreg[i][0] = vreg[i][1] = make_uint4(0, 0, 0, 0);
        if (kg < S) vreg[i][0] = *reinterpret_cast<const uint4*>(tbase + voff[i]);
        if (kg + 1 < S) vreg[i][1] = *reinterpret_cast<const uint4*>(tbase + voff[i] + rs);
      }
    }
    tbase += tile_step;
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int it = tid + 256 * i;
      const int slot = it & 7, key = (it >> 3) & 63, pl = it >> 9;
      *reinterpret_cast<uint4*>(Ks + (pl * 64 + key) * 128 + ((slot ^ (key & 7)) << 4)) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int it = tid + 256 * i;
      const int slot = it & 7, kp = (it >> 3) & 31, pl = it >> 8;
      const int key = 2 * kp;                                           // even key of the pair: key = 32 kk + 16 a + 4 gg + t, t even
      const int kk = key >> 5, a = (key >> 4) & 1, gg = (key >> 2) & 3, t = key & 3;
      const int pos = 32 * kk + 8 * gg + 4 * a + t;                      // position inside the row (bf16 units), even
      const uint32_t w0[4] = {vreg[i][0].x, vreg[i][0].y, vreg[i][0].z, vreg[i][0].w};
      const uint32_t w1[4] = {vreg[i][1].x, vreg[i][1].y, vreg[i][1].z, vreg[i][1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = 8 * slot + e;
        // (key 2 j, key 2 j + 1) of channel d as one dword: v_perm_b32 picks the two low (even e) / high (odd e) halves in ONE instruction
        const uint32_t pair = __builtin_amdgcn_perm(w1[e >> 1], w0[e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
        *reinterpret_cast<uint32_t*>(Vs + (pl * 64 + d) * 128 + (((pos >> 3) ^ (d & 7) ^ (((d >> 4) & 3) << 1)) << 4) + (pos & 7) * 2) = pair;
      }
    }
  };

  for (int kt = 0; kt < ntiles; ++kt) {
    gload(kt);                              // (no register prefetch across the tile: 48 registers; three resident blocks per CU hide the latency)
    __syncthreads();                        // every wave is done with the previous tile
    lstore();
    __syncthreads();
    if (!has_q) continue;
    // ---- S^T = K . Q^T : four 16-key fragments, six terms x two 32-deep halves each
    f32x4 sacc[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) sacc[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      uint4 kfr[3][2];
      const int key = 16 * kf + r;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) kfr[pl][kh] = *reinterpret_cast<const uint4*>(Ks + (pl * 64 + key) * 128 + (((4 * kh + g) ^ (key & 7)) << 4));
#define AT_TERM(PK, PQ)                                                                                                                   \
  _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                   \
      __builtin_bit_cast(bf16x8, kfr[PK][kh]), __builtin_bit_cast(bf16x8, qf[PQ][kh]), sacc[kf], 0, 0, 0);
      AT_TERM(0, 2) AT_TERM(2, 0) AT_TERM(1, 1) AT_TERM(0, 1) AT_TERM(1, 0) AT_TERM(0, 0)
#undef AT_TERM
    }
    // ---- online softmax on base-2 logits (lane (query r, g) holds keys 16 kf + 4 g + e)
    float mx = -INFINITY;
    const bool last = kt + 1 == ntiles;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = sacc[kf][e];                 // raw q.k: the positive scale commutes with the max and is folded into the exponent's FMA below
        if (last && kt * 64 + 16 * kf + 4 * g + e >= S) v = -INFINITY;
        sacc[kf][e] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * qscale);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    uint4 pf[3][2];                        // P as three bf16 planes, key block kk: elements t < 4 = fragment 2 kk, t >= 4 = fragment 2 kk + 1
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float pv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        pv[t] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * kk + (t >> 2)][t & 3], qscale, -m_new));
        psum += pv[t];
      }
      uint32_t hw[4], mw[4], lw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) split3_pair(pv[2 * u], pv[2 * u + 1], hw[u], mw[u], lw[u]);
      pf[0][kk] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      pf[1][kk] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
      pf[2][kk] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
    l_run = l_run * alpha + psum;
    // ---- O^T = alpha O^T + V^T . P^T : four 16-row d fragments, two 32-key blocks, six terms
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) {
      o[fd][0] *= alpha; o[fd][1] *= alpha; o[fd][2] *= alpha; o[fd][3] *= alpha;
      const int d = 16 * fd + r;
      uint4 vfr[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          vfr[pl][kk] = *reinterpret_cast<const uint4*>(Vs + (pl * 64 + d) * 128 + (((4 * kk + g) ^ (d & 7) ^ (((d >> 4) & 3) << 1)) << 4));
#define AT_TERM(PV, PP)                                                                                                                   \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                      \
      __builtin_bit_cast(bf16x8, vfr[PV][kk]), __builtin_bit_cast(bf16x8, pf[PP][kk]), o[fd], 0, 0, 0);
      AT_TERM(0, 2) AT_TERM(2, 0) AT_TERM(1, 1) AT_TERM(0, 1) AT_TERM(1, 0) AT_TERM(0, 0)
#undef AT_TERM
    }
  }
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  const int qo = q0 + r;
  if (has_q && qo < S) {
    const long at = ((long)b * S + qo) * D + h * 64 + g * 4;
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) {
      const float w4[4] = {o[fd][0] * inv, o[fd][1] * inv, o[fd][2] * inv, o[fd][3] * inv};
      store_split3(out3 + (kmaj_rows ? split3_at((long)b * S + qo, h * 64 + g * 4 + fd * 16, D, kmaj_rows) : at + fd * 16), plane_out, w4);
    }
  }
}

inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
inline int ok() { return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH; }

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int pf_patch_im2col(const float* img, int B, int H, int W, void* out, int ld, int dtype, void* stream) {
  if (!img || !out || H % 14 || W % 14 || ld < 588) return PF_ERR_ARG;
  const long total = (long)B * (H / 14) * (W / 14) * ld;
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(patch_im2col_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), img, B, H, W, (bf16_t*)out, ld);
  else hipLaunchKernelGGL(patch_im2col_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), img, B, H, W, (float*)out, ld);
  return ok();
}

extern "C" int pf_assemble_tokens(const void* emb, void* tokens, const float* cls, const float* pos, int B, int S, int D,
                                  int dtype, void* stream) {
  if (!emb || !tokens || !cls || !pos) return PF_ERR_ARG;
  const long total = (long)B * S * D;
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(assemble_tokens_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (const bf16_t*)emb, (bf16_t*)tokens, cls, pos, B, S, D);
  else hipLaunchKernelGGL(assemble_tokens_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ST(stream), (const float*)emb, (float*)tokens, cls, pos, B, S, D);
  return ok();
}

extern "C" int pf_layernorm(const void* x, int x_ld, void* y, int y_ld, const float* g, const float* b, float eps,
                            int batches, int in_rows_per_batch, int in_row_offset, int out_rows_per_batch, int D,
                            int dtype, void* stream) {
  if (!x || !y || !g || !b || D % 8 || D > 2048 || x_ld % 8 || y_ld % 8) return PF_ERR_ARG;
  const long rows = (long)batches * out_rows_per_batch;
  const int grid = (int)((rows + 3) / 4);
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(grid), dim3(256), 0, ST(stream), (const bf16_t*)x, x_ld, (bf16_t*)y, y_ld, g, b, eps, batches, in_rows_per_batch, in_row_offset, out_rows_per_batch, D);
  else hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)x, x_ld, (float*)y, y_ld, g, b, eps, batches, in_rows_per_batch, in_row_offset, out_rows_per_batch, D);
  return ok();
}

extern "C" int pf_qkv_split(const void* qkv, int B, int S, int Hh, void* q, void* k, void* vt, int Sp, float scale,
                            int dtype, void* stream) {
  if (!qkv || !q || !k || !vt || Sp % 64 || Sp < S) return PF_ERR_ARG;
  dim3 grid((S + 63) / 64, B * Hh);
  if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(qkv_split_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)qkv, B, S, Hh, (bf16_t*)q, (bf16_t*)k, (bf16_t*)vt, Sp, scale);
  else hipLaunchKernelGGL(qkv_split_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)qkv, B, S, Hh, (float*)q, (float*)k, (float*)vt, Sp, scale);
  return ok();
}

extern "C" int pf_vit_attention(const void* q, const void* k, const void* vt, void* out, int B, int S, int Sp, int Hh,
                                int dtype, void* stream) {
  if (!q || !k || !vt || !out || Sp % 64 || Sp < S) return PF_ERR_ARG;
  dim3 grid((S + 63) / 64, B * Hh);
  static int old_attn = -1;
  if (old_attn < 0) { const char* e = getenv("PF_ATTN_OLD"); old_attn = (e && e[0] == '1') ? 1 : 0; }
  if (dtype == PF_DTYPE_BF16 && !old_attn) hipLaunchKernelGGL(vit_attention32_kernel, dim3((S + 127) / 128, B * Hh), dim3(256), 0, ST(stream), (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, B, S, Sp, Hh);
  else if (dtype == PF_DTYPE_BF16) hipLaunchKernelGGL(vit_attention_kernel<bf16_t>, grid, dim3(256), 0, ST(stream), (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, B, S, Sp, Hh);
  else hipLaunchKernelGGL(vit_attention_kernel<float>, grid, dim3(256), 0, ST(stream), (const float*)q, (const float*)k, (const float*)vt, (float*)out, B, S, Sp, Hh);
  return ok();
}

static int attention_qkv(const void* qkv, void* out, long plane, int B, int S, int Hh, void* stream, long kmaj_rows = 0) {
  // head_dim^-1/2 (attention.py:55, 64^-1/2) times log2(e): the kernel's softmax runs on base-2 logits
  const float qscale = 0.125f * 1.4426950408889634f;
  // 16 queries per wave, three blocks per CU (default).  PF_ATTN_QG=2: 32 queries per wave -- the K / V^T fragments and the staging of a
  // key tile serve twice the MFMAs, but at 230 registers only two blocks fit a CU: measured 436 vs 391 us at B8 S1037 (round 3), kept
  // for A/B measurements only
  static int qg = -1;
  if (qg < 0) { const char* e = getenv("PF_ATTN_QG"); qg = (e && e[0] == '2') ? 2 : 1; }
  if (qg == 2) hipLaunchKernelGGL(vit_attention_qkv_f32_kernel<2>, dim3((S + 127) / 128, B * Hh), dim3(256), 0, ST(stream), (const float*)qkv, (float*)out, B, S, Hh, qscale, plane, kmaj_rows);
  else hipLaunchKernelGGL(vit_attention_qkv_f32_kernel<1>, dim3((S + 63) / 64, B * Hh), dim3(256), 0, ST(stream), (const float*)qkv, (float*)out, B, S, Hh, qscale, plane, kmaj_rows);
  return ok();
}

extern "C" int pf_vit_attention_qkv(const void* qkv, void* out, int B, int S, int Hh, int dtype, void* stream) {
  if (!qkv || !out || B <= 0 || S <= 0 || Hh <= 0 || dtype != PF_DTYPE_F32) return PF_ERR_ARG;
  return attention_qkv(qkv, out, 0, B, S, Hh, stream);
}

extern "C" int pf_vit_attention_qkv_split3(const void* qkv, void* out3, long plane, int kmajor, int B, int S, int Hh, void* stream) {
  if (!qkv || !out3 || B <= 0 || S <= 0 || Hh <= 0 || plane < (long)B * S * Hh * 64) return PF_ERR_ARG;
  return attention_qkv(qkv, out3, plane, B, S, Hh, stream, kmajor ? (long)B * S : 0);
}

extern "C" int pf_layernorm_split3(const float* x, int x_ld, void* y3, int y_ld, long plane, int kmajor, const float* g, const float* b, float eps,
                                   long rows, int D, void* stream) {
  if (!x || !y3 || !g || !b || D % 8 || D > 2048 || x_ld % 8 || y_ld % 8 || rows <= 0 || plane < (rows - 1) * y_ld + D) return PF_ERR_ARG;
  if (kmajor && (D % 32 || y_ld != D)) return PF_ERR_ARG;                        // chunk-major planes are dense: [D/32][rows][32]
  hipLaunchKernelGGL(layernorm_split3_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), x, x_ld, (bf16_t*)y3, y_ld, plane, g, b, eps, rows, D,
                     kmajor ? rows : 0L);
  return ok();
}

extern "C" int pf_vit_attention_split3(const void* qkv3, long plane_in, void* out3, long plane_out, int kmajor, int B, int S, int Hh, void* stream) {
  if (!qkv3 || !out3 || B <= 0 || S <= 0 || Hh <= 0 || plane_in < (long)B * S * Hh * 192 || plane_out < (long)B * S * Hh * 64) return PF_ERR_ARG;
  if (2 * plane_in + 64L * Hh * 192 + Hh * 192 >= (1L << 32)) return PF_ERR_ARG;          // (32-bit row offsets inside one 64-key tile, three planes)
  const float qscale = 0.125f * 1.4426950408889634f;        // head_dim^-1/2 (attention.py:55) times log2(e): base-2 softmax
  hipLaunchKernelGGL(vit_attention_split3_kernel, dim3((S + 63) / 64, B * Hh), dim3(256), 0, ST(stream), (const bf16_t*)qkv3, plane_in, (bf16_t*)out3,
                     plane_out, B, S, Hh, qscale, kmajor ? (long)B * S : 0L);
  return ok();
}
