// Implicit-GEMM convolution / linear layer on the CDNA4 matrix cores (gfx950).
//
//   D[n][m] = sum_k W[n][k] * X[m][k]        n = output channel, m = output pixel, k = (ky,kx,c)
//
// The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile, so that a lane's
// four accumulator registers are four CONSECUTIVE output channels of ONE pixel -> the NHWC epilogue
// store is one 8-byte (bf16) / 16-byte (f32) vector per fragment.
//
// Tiling: 256 threads = 4 waves (WM x WN); block tile BM pixels x BN channels; every K step stages
// one 128-byte row chunk per tile row (64 bf16 / 32 f32 elements) through registers into a 2-deep LDS
// ring (one barrier per chunk, next chunk's global loads issued before the MFMAs of the current one).
// LDS rows are 128 B; the 16-byte slot index is XOR-swizzled with (row>>1)&7 which makes every
// ds_read_b128 lane group of the fragment reads hit 16 distinct slots (bank-conflict free, see
// DESIGN.md "LDS swizzle").
//
// bf16: v_mfma_f32_16x16x32_bf16, lane (r=l&15, g=l>>4) holds k = g*8..g*8+7 of row r -> one 16-B slot.
// f32 : v_mfma_f32_16x16x4_f32 (exact f32 FMA chain). A lane reads one float4 at slot g; element e of
//       it feeds MFMA e, i.e. MFMA e contracts k in {4g+e}: a k-permutation applied identically to
//       both operands, which leaves the sum unchanged.
#include <atomic>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

template <typename T> __device__ __forceinline__ uint4 relu_vec(uint4 v);
template <> __device__ __forceinline__ uint4 relu_vec<float>(uint4 v) {
  v.x = (v.x & 0x80000000u) ? 0u : v.x; v.y = (v.y & 0x80000000u) ? 0u : v.y;
  v.z = (v.z & 0x80000000u) ? 0u : v.z; v.w = (v.w & 0x80000000u) ? 0u : v.w;
  return v;
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {
  uint32_t neg = ((w >> 15) & 0x00010001u) * 0xffffu;
  return w & ~neg;
}
template <> __device__ __forceinline__ uint4 relu_vec<bf16_t>(uint4 v) {
  v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w);
  return v;
}

template <typename T, int FM, int FN>
__device__ __forceinline__ void mma_half(const uint4 (&wf)[FN], const uint4 (&xf)[FM], f32x4 (&acc)[FN][FM]);

template <int FM, int FN>
__device__ __forceinline__ void mma_half_bf16(const uint4 (&wf)[FN], const uint4 (&xf)[FM], f32x4 (&acc)[FN][FM]) {
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
      acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[fn]),
                                                            __builtin_bit_cast(bf16x8, xf[fm]), acc[fn][fm], 0, 0, 0);
}
template <int FM, int FN>
__device__ __forceinline__ void mma_half_f32(const uint4 (&wf)[FN], const uint4 (&xf)[FM], f32x4 (&acc)[FN][FM]) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const uint32_t a = e == 0 ? wf[fn].x : e == 1 ? wf[fn].y : e == 2 ? wf[fn].z : wf[fn].w;
        const uint32_t b = e == 0 ? xf[fm].x : e == 1 ? xf[fm].y : e == 2 ? xf[fm].z : xf[fm].w;
        acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a), __uint_as_float(b), acc[fn][fm], 0, 0, 0);
      }
}

// (The compile-time measurement variants of rounds 1-2 -- no DMA / no LDS reads / no barrier builds, `make ablate` -- were retired in round 5 together
// with the bf16 tuning they served; their numbers are in profiles/r2_abl_f32.log.)

// 256 bytes of zeros: source of every out-of-image / out-of-range 16-byte vector (conv padding, M/N tails)
__device__ __attribute__((aligned(256))) unsigned int pf_zero_page[64];

// LDS-DMA of one 16-byte vector per lane: LDS[m0_base + lane*16] <- *gsrc.  Issued through inline asm so that
// hipcc does not serialise it against the ds_reads of the OTHER ring stage (it inserts a conservative
// s_waitcnt vmcnt(0) before the first ds_read that follows a compiler-visible LDS-DMA).  We count the
// completions ourselves: one explicit s_waitcnt vmcnt(0) before the barrier that publishes the stage.
// M0 is saved/restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// Block id -> (tile_m, tile_n) in groups of PF_IGEMM_GROUP_M pixel tiles: consecutive ids (which xcd_remap keeps
// on one XCD) walk GROUP_M pixel tiles x all channel tiles column by column, so the ~64 blocks an XCD runs
// at once form a roughly square patch of the output (8 pixel tiles x 8 channel tiles) whose A and W panels fit that
// XCD's 4 MiB L2 together, instead of 2 pixel tiles x every channel tile (the whole weight matrix streaming through L2
// once per pair of pixel tiles).
// measured on the ViT-L linears (profiles/r2_sweep_bf16_persist_shapes.log): qkv 500 -> 609, fc1 631 -> 652 TF/s with the
// 256x128 eight-wave tile; a 256x256 tile (one block per CU: 2 rounds of 396 / 3 rounds of 528 tiles, and 64 B/lane of
// epilogue spill at 256 registers) measured 576 / 558 and was dropped
#ifndef PF_PERSIST_EFF_256128
#define PF_PERSIST_EFF_256128 1.25f
#endif
#ifndef PF_F32_EFF_256      // relative efficiency of the eight-wave 256x256 f32 tile in the cost model (measured +2.2 % on 3x3 768->768 @ 8x224x296,
                            // profiles/r2_f32_tune_256.log; loses on small M, which the tail term of the model covers); 0 = forced only
#define PF_F32_EFF_256 1.03f
#endif
#ifndef PF_IGEMM_GROUP_M
#define PF_IGEMM_GROUP_M 8
#endif
__device__ __forceinline__ void tile_of(int bid, int mt, int nt, int& tile_m, int& tile_n) {
  if (PF_IGEMM_GROUP_M <= 1) {
    tile_m = bid / nt;
    tile_n = bid - tile_m * nt;
    return;
  }
  const int per_group = PF_IGEMM_GROUP_M * nt;
  const int group = bid / per_group;
  const int first_m = group * PF_IGEMM_GROUP_M;
  const int gsz = min(mt - first_m, PF_IGEMM_GROUP_M);
  const int in_g = bid - group * per_group;
  tile_n = in_g / gsz;
  tile_m = first_m + (in_g - tile_n * gsz);
}

template <typename T, int BM, int BN, int WM, int WN, bool RELU_IN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const pf_conv_params p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int BK = 8 * VEC;  // elements per 128-byte chunk row
  constexpr int NW = WM * WN;  // waves: 4 (two or more blocks per CU) or 8 (the 256x256 tile, one block per CU)
  constexpr int RP = 8 * NW;   // tile rows moved per loader pass (one 1-KiB LDS-DMA piece = 8 rows per wave)
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_ITERS = (BM + RP - 1) / RP, B_ITERS = (BN + RP - 1) / RP;
  constexpr int A_BYTES = BM * 128, B_BYTES = B_ITERS * RP * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % RP == 0, "fragment multiple");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int OHW = p.OH * p.OW;
  const int M = p.B * OHW;
  const int nt = (p.Cout + BN - 1) / BN;
  const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  int tile_m, tile_n;
  tile_of(bid, (M + BM - 1) / BM, nt, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- loader state.  One LDS-DMA instruction (global_load_lds_dwordx4) moves 1 KiB = 8 tile rows x
  // 128 B per wave: lane L lands at LDS row (L>>3), physical 16-B slot (L&7).  The XOR swizzle is applied
  // on the SOURCE side: lane L fetches logical slot j = (L&7) ^ ((row>>1)&7)  (rows advance by 32 per pass,
  // so the swizzle term is the same for every pass of a thread). ----
  const int r0 = tid >> 3;                       // tile row of pass 0
  const int j = (tid & 7) ^ ((r0 >> 1) & 7);     // logical 16-byte slot within the 128-byte chunk row
  // blockIdx.y = plane of a batched GEMM (pf_conv_params.batch: the (m+2)^2 transform points of a Winograd layer share one launch,
  // so the chip sees one stream of tiles instead of (m+2)^2 launches with a tail each); the strides are 0 / unused otherwise
  const size_t plane = blockIdx.y;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x) + plane * (size_t)p.x_bstride;
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w) + plane * (size_t)p.w_bstride;
  const char* zero = reinterpret_cast<const char*>(pf_zero_page);
  const int ntaps = p.KH * p.KW;
  const char* a_ptr[A_ITERS];
  unsigned a_mask[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int m = m0 + r0 + RP * i;
    a_mask[i] = 0u;
    a_ptr[i] = zero;
    if (m < M) {
      const int b = m / OHW, rem = m - b * OHW;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      a_ptr[i] = reinterpret_cast<const char*>(xg + ((long)b * p.H * p.W + (long)iy0 * p.W + ix0) * p.x_ld);
      unsigned mk = 0u;
      for (int t = 0; t < ntaps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W) mk |= 1u << t;
      }
      a_mask[i] = mk;
    }
  }
  const char* b_ptr[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    const int row = n0 + r0 + RP * i;
    b_ptr[i] = (row < p.w_rows) ? reinterpret_cast<const char*>(wg + (long)row * p.Kpad + j * VEC) : nullptr;
  }
  const int cin_v = p.Cin / VEC;
  const int nk = (ntaps * cin_v + 7) / 8;
  // this thread's K position: vector index kv = kc*8 + j  ->  (tap = (ky,kx), cv)
  // korder 1 (chunk-major weights): chunk kc is tap kc % ntaps of channel chunk kc / ntaps -> cv = 8 (kc / ntaps) + j
  int tap = p.korder ? 0 : j / cin_v, cv = j - tap * cin_v;
  int ky = tap / p.KW, kx = tap - ky * p.KW;

  const unsigned smem_base = lds_addr(smem);
  // GEMM fast path (1x1, K a multiple of the chunk): every source pointer just advances by 128 bytes per chunk
  // (0 for the zero page) - no tap/mask/select arithmetic in the K loop
  const bool fast = ntaps == 1 && (p.Cin % BK) == 0;
  const char* a_cur[A_ITERS];
  int a_inc[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const bool ok = a_mask[i] & 1u;
    a_cur[i] = ok ? a_ptr[i] + j * (VEC * (int)sizeof(T)) : zero;
    a_inc[i] = ok ? 128 : 0;
  }
  const char* b_cur[B_ITERS];
  int b_inc[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    b_cur[i] = b_ptr[i] ? b_ptr[i] : zero;
    b_inc[i] = b_ptr[i] ? 128 : 0;
  }
  auto issue = [&](int stage, int kc) {
    const unsigned As = smem_base + stage * STAGE + wave * (8 * 128);
    const unsigned Bs = As + A_BYTES;
    if (fast) {
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        glds16(a_cur[i], As + i * (RP * 128));
        a_cur[i] += a_inc[i];
      }
#pragma unroll
      for (int i = 0; i < B_ITERS; ++i) {
        glds16(b_cur[i], Bs + i * (RP * 128));
        b_cur[i] += b_inc[i];
      }
      return;
    }
    const long koff = ((long)(ky * p.W + kx) * p.x_ld + cv * VEC) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1u;
      const char* src = ok ? a_ptr[i] + koff : zero;
      glds16(src, As + i * (RP * 128));
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const char* src = b_ptr[i] ? b_ptr[i] + (long)kc * (BK * (long)sizeof(T)) : zero;
      glds16(src, Bs + i * (RP * 128));
    }
    // advance this thread's K position by one chunk (8 vectors)
    if (p.korder) {
      if (++kx == p.KW) { kx = 0; ++ky; }
      if (++tap == ntaps) { tap = 0; ky = 0; kx = 0; cv += 8; }
    } else {
      cv += 8;
      while (cv >= cin_v) {
        cv -= cin_v;
        ++tap;
        if (++kx == p.KW) { kx = 0; ++ky; }
      }
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addressing: lane (r, g); tile rows are multiples of 16 so the swizzle term is per-lane
  const int fr = lane & 15, fg = lane >> 4;
  // per-channel epilogue constants are fetched NOW: their global-load latency (1-2 us, once per block, a tenth of
  // a K=1024 GEMM block's life) hides behind the K loop instead of sitting between the last MFMA and the stores
  constexpr bool kPreload = FN <= 4;    // (wide tiles: 2 x FN float4 registers do not fit next to 128 accumulator registers)
  float4 bias_r[FN], scale_r[FN];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) {
    const int n = n0 + wn * WTN + fn * 16 + fg * 4;
    bias_r[fn] = make_float4(0.f, 0.f, 0.f, 0.f);
    scale_r[fn] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (kPreload && p.shuffle <= 1 && n < p.Cout) {
      if (p.bias) bias_r[fn] = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.scale) scale_r[fn] = *reinterpret_cast<const float4*>(p.scale + n);
    }
  }
  const int swz = (fr >> 1) & 7;
  const int a_row_off = (wm * WTM + fr) * 128;  // activations (pixels)
  const int b_row_off = (wn * WTN + fr) * 128;  // weights (channels)

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) issue((kc + 1) & 1, kc + 1);   // next chunk lands while this one is multiplied
    const char* As = smem + (kc & 1) * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int slot = (((s << 2) | fg) ^ swz) << 4;
      uint4 wf[FN], xf[FM];
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) wf[fn] = *reinterpret_cast<const uint4*>(Bs + b_row_off + fn * 16 * 128 + slot);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        xf[fm] = *reinterpret_cast<const uint4*>(As + a_row_off + fm * 16 * 128 + slot);
        if constexpr (RELU_IN) xf[fm] = relu_vec<T>(xf[fm]);
      }
      if constexpr (sizeof(T) == 2) mma_half_bf16<FM, FN>(wf, xf, acc);
      else mma_half_f32<FM, FN>(wf, xf, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk kc+1 has landed (this wave's pieces) ...
    __syncthreads();                                    // ... for every wave; and stage kc&1 is free again
  }

  // ---- epilogue: bias -> act -> scale -> residual(s) -> store 4 consecutive channels ----
  if constexpr (!kPreload) {
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = n0 + wn * WTN + fn * 16 + fg * 4;
      if (p.shuffle <= 1 && n < p.Cout) {
        if (p.bias) bias_r[fn] = *reinterpret_cast<const float4*>(p.bias + n);
        if (p.scale) scale_r[fn] = *reinterpret_cast<const float4*>(p.scale + n);
      }
    }
  }
  const int s = p.shuffle > 1 ? p.shuffle : 1;
  const int cout_t = p.Cout / (s * s);
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int m = m0 + wm * WTM + fm * 16 + fr;
    if (m >= M) continue;
    int b = 0, oy = 0, ox = 0;
    if (s > 1) {
      b = m / OHW;
      const int rem = m - b * OHW;
      oy = rem / p.OW;
      ox = rem - oy * p.OW;
    }
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = n0 + wn * WTN + fn * 16 + fg * 4;
      if (n >= p.Cout) continue;
      int co = n;
      long opix = m;
      if (s > 1) {
        const int q = n / cout_t;
        co = n - q * cout_t;
        const int dy = q / s, dx = q - dy * s;
        opix = ((long)b * (p.OH * s) + (oy * s + dy)) * (p.OW * s) + (ox * s + dx);
      }
      float v[4] = {acc[fn][fm][0], acc[fn][fm][1], acc[fn][fm][2], acc[fn][fm][3]};
      if (s > 1) {
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p.bias[co + r];
        }
      } else {
        v[0] += bias_r[fn].x; v[1] += bias_r[fn].y; v[2] += bias_r[fn].z; v[3] += bias_r[fn].w;
      }
      if (p.act == PF_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (p.act == PF_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (p.act == PF_ACT_SOFTPLUS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = softplus20(v[r]);
      }
      if (s > 1) {
        if (p.scale) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= p.scale[co + r];
        }
      } else {
        v[0] *= scale_r[fn].x; v[1] *= scale_r[fn].y; v[2] *= scale_r[fn].z; v[3] *= scale_r[fn].w;
      }
      if (p.res) {
        float t[4];
        load4(reinterpret_cast<const T*>(p.res) + opix * p.res_ld + co, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t[r];
      }
      if (p.res2) {
        float t[4];
        load4(reinterpret_cast<const T*>(p.res2) + opix * p.res2_ld + co, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t[r];
      }
      if (p.out_f32) store4(reinterpret_cast<float*>(p.y) + plane * (size_t)p.y_bstride + opix * p.y_ld + co, v[0], v[1], v[2], v[3]);
      else store4(reinterpret_cast<T*>(p.y) + plane * (size_t)p.y_bstride + opix * p.y_ld + co, v[0], v[1], v[2], v[3]);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Persistent variant of the 1x1 / linear fast path of conv_igemm_kernel for bf16 (default for Cout >= 2048, PF_GEMM_PERSIST=0
// turns it off, =1 / a shape code forces it for every eligible layer).  The measured life of a K=1024 ViT linear block is only
// ~55 % K loop; the rest is launch + per-block set-up, the first chunk's DMA latency and the epilogue.  Here
// 2 blocks per CU stay resident and walk the output tiles (ids b, b+G, b+2G, ... in the grouped order of tile_of);
// the K chunks of consecutive tiles form ONE stream through the two LDS stages, so the first chunk of tile t+1 is
// fetched while the last chunk of tile t is multiplied, and its second chunk while tile t's epilogue runs.
// Chunk k always lives in stage k&1 and is issued after the barrier that ends chunk k-2, exactly as in the
// one-tile kernel; `issued` counts chunks handed to the DMA engine, the cursor (it_*) is the tile/offset they
// come from.  Requirements (checked by the dispatcher): KH=KW=1, stride 1, pad 0, no shuffle, Cin % 64 == 0, Cin >= 128.
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool RELU_IN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_persist_kernel(const pf_conv_params p) {
  using T = bf16_t;
  constexpr int NT = 64 * WM * WN, RP = NT / 8;             // threads; tile rows moved per loader pass (8 per wave)
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int A_ITERS = (BM + RP - 1) / RP, B_ITERS = (BN + RP - 1) / RP;
  constexpr int A_BYTES = A_ITERS * RP * 128, B_BYTES = B_ITERS * RP * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && RP % 16 == 0, "fragment multiples; the loader swizzle needs RP % 16 == 0");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = p.B * p.OH * p.OW;
  const int mt = (M + BM - 1) / BM, nt = (p.Cout + BN - 1) / BN;
  const int T_all = mt * nt, G = (int)gridDim.x;
  const int lid = xcd_remap((int)blockIdx.x, G);          // blocks of one XCD take neighbouring tiles
  const int my_tiles = lid < T_all ? (T_all - lid + G - 1) / G : 0;
  const int nk = p.Cin / 64;
  const int total = my_tiles * nk;                        // chunks of this block's whole stream
  if (total == 0) return;

  const int r0 = tid >> 3;
  const int j = (tid & 7) ^ ((r0 >> 1) & 7);              // source-side swizzle, as in conv_igemm_kernel
  const char* __restrict__ xg = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ wg = reinterpret_cast<const char*>(p.w);
  const char* zero = reinterpret_cast<const char*>(pf_zero_page);
  const unsigned smem_base = lds_addr(smem);

  // ---- issue cursor: the tile / K offset the NEXT chunk handed to the DMA engine comes from ----
  const char* a_cur[A_ITERS];
  const char* b_cur[B_ITERS];
  int a_inc[A_ITERS], b_inc[B_ITERS];
  int it_tile = lid, it_kc = 0, issued = 0;
  auto cursor_to_tile = [&](int t) {
    int tm, tn;
    tile_of(t, mt, nt, tm, tn);
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int m = tm * BM + r0 + RP * i;
      const bool ok = (r0 + RP * i) < BM && m < M;
      a_cur[i] = ok ? xg + ((long)m * p.x_ld + j * 8) * 2 : zero;
      a_inc[i] = ok ? 128 : 0;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int row = tn * BN + r0 + RP * i;
      const bool ok = (r0 + RP * i) < BN && row < p.w_rows;
      b_cur[i] = ok ? wg + ((long)row * p.Kpad + j * 8) * 2 : zero;
      b_inc[i] = ok ? 128 : 0;
    }
  };
  auto issue_next = [&]() {
    const unsigned As = smem_base + (issued & 1) * STAGE + wave * (8 * 128);
    const unsigned Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      glds16(a_cur[i], As + i * (RP * 128));
      a_cur[i] += a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      glds16(b_cur[i], Bs + i * (RP * 128));
      b_cur[i] += b_inc[i];
    }
    ++issued;
    if (++it_kc == nk) {
      it_kc = 0;
      it_tile += G;
      if (it_tile < T_all) cursor_to_tile(it_tile);
    }
  };

  const int fr = lane & 15, fg = lane >> 4;
  const int swz = (fr >> 1) & 7;
  const int a_row_off = (wm * WTM + fr) * 128;
  const int b_row_off = (wn * WTN + fr) * 128;

  cursor_to_tile(it_tile);
  issue_next();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int c = 0;                                              // chunk being multiplied
  for (int t = lid; t < T_all; t += G) {
    int tile_m, tile_n;
    tile_of(t, mt, nt, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x4 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-channel epilogue constants: before the K loop when they are few (their latency hides behind it), after it for
    // the wide tiles (2 x FN float4 registers would not fit next to 128 accumulator registers)
    constexpr bool kPreload = FN <= 4 && FM * FN <= 16;
    float4 bias_r[FN], scale_r[FN];
    auto load_epi = [&]() {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = n0 + wn * WTN + fn * 16 + fg * 4;
        bias_r[fn] = make_float4(0.f, 0.f, 0.f, 0.f);
        scale_r[fn] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (n < p.Cout) {
          if (p.bias) bias_r[fn] = *reinterpret_cast<const float4*>(p.bias + n);
          if (p.scale) scale_r[fn] = *reinterpret_cast<const float4*>(p.scale + n);
        }
      }
    };
    if constexpr (kPreload) load_epi();
    for (int kc = 0; kc < nk; ++kc, ++c) {
      if (issued == c + 1 && issued < total) issue_next();           // keep one chunk ahead (skipped when two ahead)
      const char* As = smem + (c & 1) * STAGE;
      const char* Bs = As + A_BYTES;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int slot = (((s << 2) | fg) ^ swz) << 4;
        uint4 wf[FN], xf[FM];
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) wf[fn] = *reinterpret_cast<const uint4*>(Bs + b_row_off + fn * 16 * 128 + slot);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          xf[fm] = *reinterpret_cast<const uint4*>(As + a_row_off + fm * 16 * 128 + slot);
          if constexpr (RELU_IN) xf[fm] = relu_vec<T>(xf[fm]);
        }
        mma_half_bf16<FM, FN>(wf, xf, acc);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // tile finished: its last stage is free -> put the NEXT tile's second chunk in flight behind the epilogue
    if (issued == c + 1 && issued < total) issue_next();
    if constexpr (!kPreload) load_epi();
    // ---- epilogue (same order as conv_igemm_kernel: bias -> act -> scale -> residual(s) -> store) ----
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const int m = m0 + wm * WTM + fm * 16 + fr;
      if (m >= M) continue;
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = n0 + wn * WTN + fn * 16 + fg * 4;
        if (n >= p.Cout) continue;
        float v[4] = {acc[fn][fm][0] + bias_r[fn].x, acc[fn][fm][1] + bias_r[fn].y, acc[fn][fm][2] + bias_r[fn].z,
                      acc[fn][fm][3] + bias_r[fn].w};
        if (p.act == PF_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (p.act == PF_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        } else if (p.act == PF_ACT_SOFTPLUS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = softplus20(v[r]);
        }
        v[0] *= scale_r[fn].x; v[1] *= scale_r[fn].y; v[2] *= scale_r[fn].z; v[3] *= scale_r[fn].w;
        if (p.res) {
          float r4[4];
          load4(reinterpret_cast<const T*>(p.res) + (long)m * p.res_ld + n, r4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += r4[r];
        }
        if (p.res2) {
          float r4[4];
          load4(reinterpret_cast<const T*>(p.res2) + (long)m * p.res2_ld + n, r4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += r4[r];
        }
        if (p.out_f32) store4(reinterpret_cast<float*>(p.y) + (long)m * p.y_ld + n, v[0], v[1], v[2], v[3]);
        else store4(reinterpret_cast<T*>(p.y) + (long)m * p.y_ld + n, v[0], v[1], v[2], v[3]);
      }
    }
  }
}


// ---- launch helpers.  The >64 KiB dynamic-LDS opt-in is a per-DEVICE function attribute: one bit per device ordinal,
// set atomically, so a second GPU in the same process (or a second host thread) gets its own hipFuncSetAttribute.
// The launch status is captured ONCE (hipGetLastError clears it) and kept for pf_conv's error message. ----
thread_local hipError_t g_launch_status = hipSuccess;
inline void ensure_dynamic_lds(const void* kern, int smem, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    done.fetch_or(bit, std::memory_order_release);
  }
}
inline int launch_status() {
  g_launch_status = hipGetLastError();
  return g_launch_status == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

// candidate shapes of the persistent GEMM: {BM, BN, WM, WN}; resident blocks per CU follow from the 160 KiB of LDS
template <int BM, int BN, int WM, int WN>
struct PersistCfg {
  static constexpr int RP = 8 * WM * WN;
  static constexpr int smem = 2 * (((BM + RP - 1) / RP) * RP + ((BN + RP - 1) / RP) * RP) * 128;
  static constexpr int occ = (160 * 1024) / smem;
};

template <int BM, int BN, int WM, int WN, bool RELU_IN>
int launch_persist(const pf_conv_params& p, hipStream_t st) {
  using C = PersistCfg<BM, BN, WM, WN>;
  constexpr int smem = C::smem;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = gemm_persist_kernel<BM, BN, WM, WN, RELU_IN>;
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  const long M = (long)p.B * p.OH * p.OW;
  const long tiles = ((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  const long slots = 256L * C::occ;                      // resident blocks
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(64 * WM * WN), smem, st, p);
  return launch_status();
}

// -------------------------------------------------------------------------------------------------
// "big" bf16 variant for the layers that carry the FLOPs (Cout >= 96, many pixels):
// 512 threads = 8 waves (4 along pixels x 2 along channels), block tile 256 pixels x 128 channels, per-wave
// 64x64 from 2x2 v_mfma_f32_32x32x16_bf16 fragments, K chunks of 64 through a THREE-deep LDS-DMA ring
// (3 x 48 KiB = 144 KiB, one block per CU, two waves per SIMD).  Chunk k+2 is issued before chunk k is
// multiplied and the per-chunk wait is a COUNTED s_waitcnt vmcnt(6) (= the 6 DMA instructions of the newest
// chunk stay in flight across the barrier), so HBM/L2 latency is covered by two chunks of MFMA work.
// Same LDS image / source-side swizzle / zero-page halo as conv_igemm_kernel.
// 32x32x16 operand layout: lane l holds row (l&31), k = (l>>5)*8..+7 of a 16-deep step -> 16-byte slot
// 2t + (l>>5) of the 128-byte row; D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// -------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool RELU_IN>
__global__ __launch_bounds__(512) void conv_igemm_big_kernel(const pf_conv_params p) {
  using T = bf16_t;
  constexpr int VEC = 8, BK = 64;
  constexpr int BM = 256, BN = 128, NST = 3;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_ITERS = BM / 64, B_ITERS = BN / 64;  // 512 threads = 64 rows x 8 slots per pass
  constexpr int NDMA = A_ITERS + B_ITERS;              // DMA instructions per wave per chunk (6)
  static_assert(NDMA == 6, "vmcnt immediate below assumes 6");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int OHW = p.OH * p.OW;
  const int M = p.B * OHW;
  const int nt = (p.Cout + BN - 1) / BN;
  const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  int tile_m, tile_n;
  tile_of(bid, (M + BM - 1) / BM, nt, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int r0 = tid >> 3;                       // 0..63
  const int j = (tid & 7) ^ ((r0 >> 1) & 7);     // logical slot fetched by this lane (source-side swizzle)
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const char* zero = reinterpret_cast<const char*>(pf_zero_page);
  const int ntaps = p.KH * p.KW;
  const char* a_ptr[A_ITERS];
  unsigned a_mask[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int m = m0 + r0 + 64 * i;
    a_mask[i] = 0u;
    a_ptr[i] = zero;
    if (m < M) {
      const int b = m / OHW, rem = m - b * OHW;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      a_ptr[i] = reinterpret_cast<const char*>(xg + ((long)b * p.H * p.W + (long)iy0 * p.W + ix0) * p.x_ld);
      unsigned mk = 0u;
      for (int t = 0; t < ntaps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W) mk |= 1u << t;
      }
      a_mask[i] = mk;
    }
  }
  const char* b_ptr[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    const int row = n0 + r0 + 64 * i;
    b_ptr[i] = (row < p.w_rows) ? reinterpret_cast<const char*>(wg + (long)row * p.Kpad + j * VEC) : nullptr;
  }
  const int cin_v = p.Cin / VEC;
  const int nk = (ntaps * cin_v + 7) / 8;
  int tap = j / cin_v, cv = j - tap * cin_v;
  int ky = tap / p.KW, kx = tap - ky * p.KW;

  const unsigned smem_base = lds_addr(smem);
  auto issue = [&](int stage, int kc) {
    const unsigned As = smem_base + stage * STAGE + wave * (8 * 128);
    const unsigned Bs = As + A_BYTES;
    const long koff = ((long)(ky * p.W + kx) * p.x_ld + cv * VEC) * (long)sizeof(T);
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1u;
      const char* src = ok ? a_ptr[i] + koff : zero;
      glds16(src, As + i * (64 * 128));
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const char* src = b_ptr[i] ? b_ptr[i] + (long)kc * (BK * (long)sizeof(T)) : zero;
      glds16(src, Bs + i * (64 * 128));
    }
    cv += 8;
    while (cv >= cin_v) {
      cv -= cin_v;
      ++tap;
      if (++kx == p.KW) { kx = 0; ++ky; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int fn = 0; fn < 2; ++fn)
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fn][fm][e] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int swz = (fr >> 1) & 7;
  const int a_row_off = (wm * 64 + fr) * 128;  // activations (pixels)
  const int b_row_off = (wn * 64 + fr) * 128;  // weights (channels)

  issue(0, 0);
  if (nk > 1) {
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int st = 0;       // ring stage of chunk kc
  for (int kc = 0; kc < nk; ++kc) {
    const bool more2 = kc + 2 < nk;
    if (more2) issue(st >= 1 ? st - 1 : 2, kc + 2);   // (st + 2) % 3 : the stage consumed in iteration kc-1
    const char* As = smem + st * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int slot = (((t << 1) | fh) ^ swz) << 4;
      uint4 wf[2], xf[2];
#pragma unroll
      for (int fn = 0; fn < 2; ++fn) wf[fn] = *reinterpret_cast<const uint4*>(Bs + b_row_off + fn * 32 * 128 + slot);
#pragma unroll
      for (int fm = 0; fm < 2; ++fm) {
        xf[fm] = *reinterpret_cast<const uint4*>(As + a_row_off + fm * 32 * 128 + slot);
        if constexpr (RELU_IN) xf[fm] = relu_vec<T>(xf[fm]);
      }
#pragma unroll
      for (int fn = 0; fn < 2; ++fn)
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
          acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[fn]),
                                                                __builtin_bit_cast(bf16x8, xf[fm]), acc[fn][fm], 0, 0, 0);
    }
    if (more2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // chunk kc+1 landed, kc+2 stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    st = st == 2 ? 0 : st + 1;
  }

  // ---- epilogue (same fused form as conv_igemm_kernel) ----
  const int s = p.shuffle > 1 ? p.shuffle : 1;
  const int cout_t = p.Cout / (s * s);
#pragma unroll
  for (int fm = 0; fm < 2; ++fm) {
    const int m = m0 + wm * 64 + fm * 32 + fr;
    if (m >= M) continue;
    int b = 0, oy = 0, ox = 0;
    if (s > 1) {
      b = m / OHW;
      const int rem = m - b * OHW;
      oy = rem / p.OW;
      ox = rem - oy * p.OW;
    }
#pragma unroll
    for (int fn = 0; fn < 2; ++fn) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + fn * 32 + 8 * q + 4 * fh;
        if (n >= p.Cout) continue;
        int co = n;
        long opix = m;
        if (s > 1) {
          const int qq = n / cout_t;
          co = n - qq * cout_t;
          const int dy = qq / s, dx = qq - dy * s;
          opix = ((long)b * (p.OH * s) + (oy * s + dy)) * (p.OW * s) + (ox * s + dx);
        }
        float v[4] = {acc[fn][fm][4 * q], acc[fn][fm][4 * q + 1], acc[fn][fm][4 * q + 2], acc[fn][fm][4 * q + 3]};
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p.bias[co + r];
        }
        if (p.act == PF_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (p.act == PF_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        } else if (p.act == PF_ACT_SOFTPLUS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = softplus20(v[r]);
        }
        if (p.scale) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= p.scale[co + r];
        }
        if (p.res) {
          float t4[4];
          load4(reinterpret_cast<const T*>(p.res) + opix * p.res_ld + co, t4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += t4[r];
        }
        if (p.res2) {
          float t4[4];
          load4(reinterpret_cast<const T*>(p.res2) + opix * p.res2_ld + co, t4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += t4[r];
        }
        if (p.out_f32) store4(reinterpret_cast<float*>(p.y) + opix * p.y_ld + co, v[0], v[1], v[2], v[3]);
        else store4(reinterpret_cast<T*>(p.y) + opix * p.y_ld + co, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <bool RELU_IN>
int launch_big(const pf_conv_params& p, hipStream_t st) {
  constexpr int smem = 3 * (256 + 128) * 128;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = conv_igemm_big_kernel<RELU_IN>;
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  const long M = (long)p.B * p.OH * p.OW;
  const long mt = (M + 255) / 256, nt = (p.Cout + 127) / 128;
  hipLaunchKernelGGL(kern, dim3((unsigned)(mt * nt)), dim3(512), smem, st, p);
  return launch_status();
}

// -------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with TAP REUSE (the FLOP-dominant class: fusion U-Net, DPT RCUs):
// the implicit-GEMM kernels above re-fetch every input pixel once per filter tap (9x) through the
// per-CU global->LDS path, which ablation shows to be as expensive as the MFMA work itself.  Here a block
// owns a 16x32-pixel output tile of one image; per 32-channel chunk the 18x34 input HALO tile is brought
// into LDS ONCE (double buffered) and all nine taps are multiplied out of it, only the 8-12 KiB weight
// tile changes per tap (3-deep ring).  8 waves = 4 (pixel rows) x 2 (channels); wave tile 128 px x 32*FN
// channels from 4 x FN v_mfma_f32_32x32x16_bf16 fragments (FN = 2 -> BN 128, FN = 3 -> BN 192).
// LDS rows are 64 B (32 bf16 channels); 16-byte slot swizzle phys = slot ^ ((row>>2)&3) is conflict free for
// any 32-row fragment start (halo fragments start at arbitrary rows) -- DESIGN.md section 4.
// A step = one filter ROW (3 taps) of one chunk: its 3 x BN x 64 B weight tiles sit in a 2-deep ring; at the
// start of step g every wave issues a third of the NEXT chunk's halo and the weights of step g+1, which have
// the whole step (72 MFMAs per wave) to land before the single vmcnt(0)+barrier that ends it.
// -------------------------------------------------------------------------------------------------
// ---- hand-counted LDS reads (see the 12-fragment branch of conv3x3_halo_kernel).  An asm ds_read is invisible to
// hipcc's s_waitcnt bookkeeping; lds_wait<N> names the registers whose data it guarantees so that no consumer is
// scheduled above it.  (native vector type: HIP's uint4 is a struct and cannot be a tied asm register operand) ----
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
__device__ __forceinline__ u32x4 lds_read16_asm(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16_asm_off(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& a) {
  u32x4 ta = a;
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ta) : "n"(N));
  a = ta;
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  u32x4 ta = a, tb = b, tc = c, td = d;
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ta), "+v"(tb), "+v"(tc), "+v"(td) : "n"(N));
  a = ta; b = tb; c = tc; d = td;
}

template <int N>
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f) {
  u32x4 ta = a, tb = b, tc = c, td = d, te = e, tf = f;
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(ta), "+v"(tb), "+v"(tc), "+v"(td), "+v"(te), "+v"(tf) : "n"(N));
  a = ta; b = tb; c = tc; d = td; e = te; f = tf;
}

constexpr bool kHaloRoll = true;

template <int WN, int FM, int FN, bool RELU_IN>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const pf_conv_params p) {
  using T = bf16_t;
  static_assert((8 / WN) * FM == 16, "8 waves cover the 16 tile rows");
  constexpr int TH = 16, TW = 32, HW_ = TW + 2, HROWS = (TH + 2) * HW_;   // 612 halo pixels
  constexpr int A_PIECES = (HROWS + 15) / 16;                               // 39 DMA pieces of 16 rows
  constexpr int A_PER_STEP = (A_PIECES + 2) / 3;                            // 13 pieces issued per step (3 steps / chunk)
  constexpr int A_BUF = 40960;                                              // >= 39*16*64
  constexpr int BN = WN * 32 * FN, W_TILE = BN * 64, W_PIECES = BN / 16;    // per-tap weight tile: BN rows x 64 B
  constexpr int W_STAGE = 3 * W_TILE, WQ = 3 * W_PIECES;                    // one filter ROW (3 taps) per step
  constexpr int WQ_PER_WAVE = (WQ + 7) / 8;                                 // 3 (BN 128) or 5 (BN 192)
  constexpr int LDS_W0 = 2 * A_BUF;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> (pixel-row group wm, channel group wn).  Waves w and w+4 share a SIMD; for WN = 2 they get different
  // channel halves and row groups two apart (measured +1 % over wm = wave / 2, wn = wave % 2 on the 544->544 layer).
  const int wm = WN == 2 ? (wave < 4 ? wave : ((wave + 2) & 3)) : wave / WN;
  const int wn = WN == 2 ? (wave >> 2) : wave % WN;
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
  const int nt = (p.Cout + BN - 1) / BN;
  const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = bid / nt, tile_n = bid - tile_m * nt;
  const int b = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - b * tiles_x * tiles_y;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const int n0 = tile_n * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const char* zero = reinterpret_cast<const char*>(pf_zero_page);

  // ---- DMA issue (LOADER waves only).  The two waves that share a SIMD are waves w and w+4; only waves 0..3
  // issue the LDS-DMAs of a step, so while they spend ~1000 cycles on address arithmetic + DMA issue their
  // partners 4..7 already run MFMAs: the matrix pipe is never idle because BOTH co-resident waves are in a
  // non-MFMA phase (measured: SQ_WAIT_ANY 41 % with symmetric issue).
  // A piece = 16 halo rows x 64 B: lane L -> row 16*piece + (L>>2), physical slot L&3, fetching logical slot
  // (L&3) ^ ((row>>2)&3).  Step ky of a chunk brings pieces [13 ky, 13 ky + 13) of the NEXT chunk's halo. ----
  const bool loader = wave < 4;
  const char* xbase = reinterpret_cast<const char*>(xg);
  const char* wbase = reinterpret_cast<const char*>(wg);
  const int l_row = lane >> 2, l_slot = lane & 3;
  const int nchunks = p.Cin / 32;
  const int G = nchunks * 3;                 // steps: (chunk, ky)
  const unsigned smem_base = lds_addr(smem);

  auto issue_a = [&](int st, int chunk) {    // halo pieces of step-slot st for channel chunk `chunk`
#pragma unroll
    for (int i = 0; i < (A_PER_STEP + 3) / 4; ++i) {
      const int q = wave + 4 * i;
      const int piece = st * A_PER_STEP + q;
      if (q < A_PER_STEP && piece < A_PIECES) {
        const int row = piece * 16 + l_row;
        const int jl = l_slot ^ ((row >> 2) & 3);
        const int hy = row / HW_, hx = row - hy * HW_;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = row < HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const char* src = ok ? xbase + ((((long)b * p.H + iy) * p.W + ix) * p.x_ld + jl * 8 + chunk * 32) * 2 : zero;
        glds16(src, smem_base + (chunk & 1) * A_BUF + piece * 1024);
      }
    }
  };
  auto issue_w = [&](int g) {                // weight row-tile of step g = chunk*3 + ky -> ring slot g&1
    const int chunk = g / 3, kyy = g - chunk * 3;
    const long koff = (long)(kyy * 3) * p.Cin + chunk * 32;
    const unsigned dst = smem_base + LDS_W0 + (g & 1) * W_STAGE;
#pragma unroll
    for (int i = 0; i < (WQ + 3) / 4; ++i) {
      const int q = wave + 4 * i;
      if (q < WQ) {
        const int kxq = q / W_PIECES, piece = q - kxq * W_PIECES;
        const int row = piece * 16 + l_row;
        const int jl = l_slot ^ ((row >> 2) & 3);
        const bool ok = (n0 + row) < p.w_rows;
        const char* src = ok ? wbase + ((long)(n0 + row) * p.Kpad + koff + (long)kxq * p.Cin + jl * 8) * 2 : zero;
        glds16(src, dst + q * 1024);
      }
    }
  };

  f32x16 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fn][fm][e] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  int arow_base[FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) arow_base[fm] = (FM * wm + fm) * HW_ + fr;
  int w_off[FN], w_swz[FN];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) {
    const int row = wn * (32 * FN) + fn * 32 + fr;
    w_off[fn] = row * 64;
    w_swz[fn] = (row >> 2) & 3;
  }

  unsigned roll_w[2];                        // (hand-scheduled 12-fragment path) weight read address per k-half, ring slot 0
  {
    const int o0 = (fh ^ ((fr >> 2) & 3)) << 4;
    roll_w[0] = smem_base + LDS_W0 + (wn * (32 * FN) + fr) * 64 + o0;
    roll_w[1] = smem_base + LDS_W0 + (wn * (32 * FN) + fr) * 64 + (o0 ^ 32);
  }

  // ---- prologue: whole halo of chunk 0 and the weight rows of step 0 ----
  if (loader) {
    issue_a(0, 0);
    issue_a(1, 0);
    issue_a(2, 0);
    issue_w(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int chunk = 0, ky = 0;
  for (int g = 0; g < G; ++g) {
    // ---- loader waves: a third of the next chunk's halo, then the weight rows of step g+1 (other ring slot) ----
    if (loader) {
      if (chunk + 1 < nchunks) issue_a(ky, chunk + 1);
      if (g + 1 < G) issue_w(g + 1);
    }
    // ---- multiply filter row ky (three taps) of this chunk ----
    const char* Ab = smem + (chunk & 1) * A_BUF;
    const char* Wb = smem + LDS_W0 + (g & 1) * W_STAGE;
    if constexpr (kHaloRoll && FN == 2 && FM == 4) {
      // 8-fragment tile (128 accumulator registers): room for two complete fragment sets.  Same hand-counted asm
      // reads as the 12-fragment path below: the six reads of sub-step u+1 are issued BEFORE the eight MFMAs of
      // sub-step u (hipcc, left to itself, sinks them next to their first use and waits lgkmcnt(0) in between).
      const unsigned a_lds = smem_base + (chunk & 1) * A_BUF;
      u32x4 wf[2][FN], xf[2][FM];
      auto rd_all = [&](auto uc) {
        constexpr int u = decltype(uc)::value, kx = u >> 1, t = u & 1, bsel = u & 1;
        wf[bsel][0] = lds_read16_asm_off<kx * W_TILE>(roll_w[t]);
        wf[bsel][1] = lds_read16_asm_off<kx * W_TILE + 2048>(roll_w[t]);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          const int row = fr + (FM * wm + fm + ky) * HW_ + kx;
          xf[bsel][fm] = lds_read16_asm(a_lds + row * 64 + ((((t << 1) | fh) ^ ((row >> 2) & 3)) << 4));
        }
      };
      auto substep = [&](auto uc) {
        constexpr int u = decltype(uc)::value, c = u & 1;
        constexpr bool more = u + 1 < 6;
        if constexpr (more) {
          rd_all(std::integral_constant<int, (more ? u + 1 : u)>{});
          lds_wait<6>(wf[c][0], wf[c][1], xf[c][0], xf[c][1], xf[c][2], xf[c][3]);
        } else {
          lds_wait<0>(wf[c][0], wf[c][1], xf[c][0], xf[c][1], xf[c][2], xf[c][3]);
        }
        if constexpr (RELU_IN) {
#pragma unroll
          for (int fm = 0; fm < FM; ++fm) {
            uint4 v = __builtin_bit_cast(uint4, xf[c][fm]);
            xf[c][fm] = __builtin_bit_cast(u32x4, relu_vec<T>(v));
          }
        }
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
            acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c][fn]),
                                                                  __builtin_bit_cast(bf16x8, xf[c][fm]), acc[fn][fm], 0, 0, 0);
      };
      rd_all(std::integral_constant<int, 0>{});
      substep(std::integral_constant<int, 0>{}); substep(std::integral_constant<int, 1>{});
      substep(std::integral_constant<int, 2>{}); substep(std::integral_constant<int, 3>{});
      substep(std::integral_constant<int, 4>{}); substep(std::integral_constant<int, 5>{});
      {
        const int d = (g & 1) ? -W_STAGE : W_STAGE;
        roll_w[0] += d; roll_w[1] += d;
      }
    } else if constexpr (FN * FM <= 8) {
      // software-pipelined fragment reads (enough registers when the accumulator tile is <= 8 fragments):
      // the ds_reads of sub-step u+1 = (kx, t) are in flight while the MFMAs of sub-step u execute
      uint4 wf[2][FN], xf[2][FM];
      auto load_frags = [&](int u, uint4 (&w)[FN], uint4 (&x)[FM]) {
        const int kx = u >> 1, t = u & 1;
        const int tap_off = ky * HW_ + kx;
        const int slot = (t << 1) | fh;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          w[fn] = *reinterpret_cast<const uint4*>(Wb + kx * W_TILE + w_off[fn] + ((slot ^ w_swz[fn]) << 4));
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          const int row = arow_base[fm] + tap_off;
          x[fm] = *reinterpret_cast<const uint4*>(Ab + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
          if constexpr (RELU_IN) x[fm] = relu_vec<T>(x[fm]);
        }
      };
      load_frags(0, wf[0], xf[0]);
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (u + 1 < 6) load_frags(u + 1, wf[(u + 1) & 1], xf[(u + 1) & 1]);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
            acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[u & 1][fn]),
                                                                  __builtin_bit_cast(bf16x8, xf[u & 1][fm]), acc[fn][fm], 0, 0, 0);
      }
    } else if constexpr (kHaloRoll && !RELU_IN) {
      // 12-fragment tile: 192 of the wave's 256 registers hold accumulators, so a second fragment set does not fit,
      // and hipcc sinks every ds_read next to its first use (`ds_read; s_waitcnt lgkmcnt(0); mfma`: the full LDS
      // latency exposed ~20x per step).  Hand-placed rolling prefetch instead: LDS reads are inline asm (invisible to
      // hipcc's waitcnt bookkeeping), the pixel fragment is the OUTER loop, x[fm] is dead after its FN MFMAs and the
      // read of the NEXT sub-step's x[fm] goes into the same registers right there; the next w[fn] is read right after
      // the last MFMA of the sub-step that uses w[fn].  LDS returns in order, so each wait is a fixed count of the
      // younger reads that may still be in flight; every wait names the registers it releases so that the MFMAs
      // stay below it.  Weight addresses: the swizzle term (row>>2)&3 of row = 96 wn + 32 fn + fr is (fr>>2)&3 for
      // every fn, so ONE register per k-half (roll_w[t], advanced between the two ring slots at the end of the step)
      // addresses all of them through the 16-bit offset field (tap kx: + kx * W_TILE, fragment fn: + 2048 fn).
      static_assert(FN == 3 && FM == 4, "issue order and wait counts below are for the 3x4 fragment tile");
      const unsigned a_lds = smem_base + (chunk & 1) * A_BUF;
      u32x4 wf[FN], xf[FM];
      auto rd_x = [&](auto uc, auto fmc) {
        constexpr int u = decltype(uc)::value, fm = decltype(fmc)::value;
        constexpr int kx = u >> 1, t = u & 1;
        const int row = fr + (FM * wm + fm + ky) * HW_ + kx;        // uniform part lives in SGPRs
        const unsigned ad = a_lds + row * 64 + ((((t << 1) | fh) ^ ((row >> 2) & 3)) << 4);
        xf[fm] = lds_read16_asm(ad);
      };
      auto rd_w = [&](auto uc, auto fnc) {
        constexpr int u = decltype(uc)::value, fn = decltype(fnc)::value;
        wf[fn] = lds_read16_asm_off<(u >> 1) * W_TILE + fn * 2048>(roll_w[u & 1]);
      };
      auto mma = [&](int fn, int fm) {
        acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[fn]),
                                                              __builtin_bit_cast(bf16x8, xf[fm]), acc[fn][fm], 0, 0, 0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      // LDS issue order per sub-step: x0 x1 x2 | w0 w1 w2 | x3
      rd_x(I0{}, I0{}); rd_x(I0{}, I1{}); rd_x(I0{}, I2{});
      rd_w(I0{}, I0{}); rd_w(I0{}, I1{}); rd_w(I0{}, I2{});
      rd_x(I0{}, I3{});
      auto substep = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr bool more = u + 1 < 6;
        using UN = std::integral_constant<int, (more ? u + 1 : u)>;
        // younger than w0: w1 w2 x3 -> 3, than w1 -> 2, than w2 -> 1 (x0..x2 are older than the w's)
        lds_wait<3>(wf[0], xf[0], xf[1], xf[2]);
        mma(0, 0);
        lds_wait<2>(wf[1]);
        mma(1, 0);
        lds_wait<1>(wf[2]);
        mma(2, 0);
        if constexpr (more) rd_x(UN{}, I0{});
        mma(0, 1); mma(1, 1); mma(2, 1);
        if constexpr (more) rd_x(UN{}, I1{});
        mma(0, 2); mma(1, 2); mma(2, 2);
        if constexpr (more) rd_x(UN{}, I2{});
        if constexpr (more) lds_wait<3>(xf[3]); else lds_wait<0>(xf[3]);   // younger than x3: the three x reads just issued
        mma(0, 3);
        if constexpr (more) rd_w(UN{}, I0{});
        mma(1, 3);
        if constexpr (more) rd_w(UN{}, I1{});
        mma(2, 3);
        if constexpr (more) { rd_w(UN{}, I2{}); rd_x(UN{}, I3{}); }
      };
      substep(I0{}); substep(I1{}); substep(I2{}); substep(I3{});
      substep(std::integral_constant<int, 4>{}); substep(std::integral_constant<int, 5>{});
      {
        const int d = (g & 1) ? -W_STAGE : W_STAGE;                        // other ring slot for the next step
        roll_w[0] += d; roll_w[1] += d;
      }
    } else {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tap_off = ky * HW_ + kx;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int slot = (t << 1) | fh;
        uint4 wf[FN], xf[FM];
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          wf[fn] = *reinterpret_cast<const uint4*>(Wb + kx * W_TILE + w_off[fn] + ((slot ^ w_swz[fn]) << 4));
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          const int row = arow_base[fm] + tap_off;
          xf[fm] = *reinterpret_cast<const uint4*>(Ab + row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
          if constexpr (RELU_IN) xf[fm] = relu_vec<T>(xf[fm]);
        }
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
            acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[fn]),
                                                                  __builtin_bit_cast(bf16x8, xf[fm]), acc[fn][fm], 0, 0, 0);
      }
    }
    }
    // ---- everything issued this step (weights of g+1, halo pieces) has a whole step of MFMA to land ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (++ky == 3) { ky = 0; ++chunk; }
  }

  // ---- epilogue ----
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int oy = ty0 + FM * wm + fm, ox = tx0 + fr;
    if (oy >= p.H || ox >= p.W) continue;
    const long opix = ((long)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + wn * (32 * FN) + fn * 32 + 8 * q + 4 * fh;
        if (co >= p.Cout) continue;
        float v[4] = {acc[fn][fm][4 * q], acc[fn][fm][4 * q + 1], acc[fn][fm][4 * q + 2], acc[fn][fm][4 * q + 3]};
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p.bias[co + r];
        }
        if (p.act == PF_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (p.act == PF_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        } else if (p.act == PF_ACT_SOFTPLUS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = softplus20(v[r]);
        }
        if (p.scale) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= p.scale[co + r];
        }
        if (p.res) {
          float t4[4];
          load4(reinterpret_cast<const T*>(p.res) + opix * p.res_ld + co, t4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += t4[r];
        }
        if (p.res2) {
          float t4[4];
          load4(reinterpret_cast<const T*>(p.res2) + opix * p.res2_ld + co, t4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += t4[r];
        }
        if (p.out_f32) store4(reinterpret_cast<float*>(p.y) + opix * p.y_ld + co, v[0], v[1], v[2], v[3]);
        else store4(reinterpret_cast<T*>(p.y) + opix * p.y_ld + co, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <int WN, int FM, int FN, bool RELU_IN>
int launch_halo(const pf_conv_params& p, hipStream_t st) {
  constexpr int BN = WN * 32 * FN;
  constexpr int smem = 2 * 40960 + 2 * 3 * BN * 64;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = conv3x3_halo_kernel<WN, FM, FN, RELU_IN>;
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  const long tiles = (long)p.B * ((p.H + 15) / 16) * ((p.W + 31) / 32);
  const long nt = (p.Cout + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * nt)), dim3(512), smem, st, p);
  return launch_status();
}

thread_local char g_err[256] = {0};

template <typename T, int BM, int BN, int WM, int WN, bool RELU_IN>
int launch_cfg2(const pf_conv_params& p, hipStream_t st) {
  constexpr int RP = 8 * WM * WN;
  constexpr int smem = 2 * (BM + ((BN + RP - 1) / RP) * RP) * 128;
  static std::atomic<unsigned long long> attr_done{0};
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, RELU_IN>;
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, attr_done);
  const long M = (long)p.B * p.OH * p.OW;
  const long mt = (M + BM - 1) / BM, nt = (p.Cout + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3((unsigned)(mt * nt), (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(64 * WM * WN), smem, st, p);
  return launch_status();
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const pf_conv_params& p, hipStream_t st) {
  return p.relu_in ? launch_cfg2<T, BM, BN, WM, WN, true>(p, st) : launch_cfg2<T, BM, BN, WM, WN, false>(p, st);
}

int g_force_small = -1;   // PF_IGEMM_SMALL=1 forces the 4-wave kernel everywhere (A/B measurements)

// Tile selection for the generic implicit-GEMM kernels: a wave-quantisation-aware cost model.
// A config with block tile BMxBN that admits `occ` blocks per CU processes 256*occ blocks per "round";
// each round costs ~ BM*BN*occ (all configs sustain a similar per-CU rate), scaled by a small
// efficiency factor for the narrow tiles.  At the ViT sizes (8296 tokens x 1024 channels) this is the
// difference between 2 half-empty rounds of 256x128 tiles and 1 full round of 128x96 tiles.
template <typename T> int dispatch_generic(const pf_conv_params& p, hipStream_t st, bool allow_split);

static bool force_halo() {
  const char* e = getenv("PF_HALO_FORCE");
  return e && e[0] == '1';
}

template <typename T>
int dispatch(const pf_conv_params& p, hipStream_t st) {
  if (g_force_small < 0) {
    const char* e = getenv("PF_IGEMM_SMALL");
    g_force_small = e ? atoi(e) : 0;   // 1: 4-wave kernel everywhere; 2: no halo kernel (big/small only)
  }
  const long M = (long)p.B * p.OH * p.OW;
  if constexpr (sizeof(T) == 2) {
    if (g_force_small != 1 && g_force_small != 2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.shuffle <= 1 &&
        p.Cin % 32 == 0 && p.H >= 16 && p.W >= 32 &&
        // enough 16x32 tiles to fill the chip four times over; on small maps the padded tiles (e.g. 56x74 -> 64x96)
        // and the few long-running blocks lose to the generic kernel (measured at L0..L3 of the pyramid).
        // PF_HALO_FORCE=1 (read per call; unit tests) routes small shapes here too.
        (force_halo() || (M >= 2048 && (long)p.B * ((p.H + 15) / 16) * ((p.W + 31) / 32) * ((p.Cout + 127) / 128) >= 1024))) {
      if (p.Cout <= 32) return p.relu_in ? launch_halo<1, 2, 1, true>(p, st) : launch_halo<1, 2, 1, false>(p, st);
      if (p.Cout <= 64) return p.relu_in ? launch_halo<1, 2, 2, true>(p, st) : launch_halo<1, 2, 2, false>(p, st);
      // channel tile: 192 when it wastes less than 128 (e.g. 544 -> 3x192 = 576 vs 5x128 = 640; 768 -> 4x192)
      const int pad128 = (p.Cout + 127) / 128 * 128, pad192 = (p.Cout + 191) / 192 * 192;
      static int fn_force = -1;
      if (fn_force < 0) { const char* e = getenv("PF_HALO_FN"); fn_force = e ? atoi(e) : 0; }
      if (fn_force == 2) return p.relu_in ? launch_halo<2, 4, 2, true>(p, st) : launch_halo<2, 4, 2, false>(p, st);
      if (pad192 <= pad128) return p.relu_in ? launch_halo<2, 4, 3, true>(p, st) : launch_halo<2, 4, 3, false>(p, st);
      return p.relu_in ? launch_halo<2, 4, 2, true>(p, st) : launch_halo<2, 4, 2, false>(p, st);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // persistent GEMM (see gemm_persist_kernel; validated by tests/test_persistent_gemm_gpu.py).
    // PF_GEMM_PERSIST (read per call): 0 = off, 1 = every eligible layer with the shape picked by the makespan model below,
    // or force "BMxBN" by its code 128128 / 12896 / 12864 / 144128 / 14464 / 256128.
    const char* pe = getenv("PF_GEMM_PERSIST");
    // default: on for the wide linears (Cout >= 2048: ViT qkv / fc1, +2 ... +16 % measured, profiles/r2_sweep_bf16*.log);
    // the narrow ones (proj, fc2: 520 tiles on 512 resident blocks) measured +-2 % and keep the one-tile kernel
    const int persist = pe ? atoi(pe) : (p.Cout >= 2048 ? 1 : 0);
    if (persist > 0 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.shuffle <= 1 && p.Cin % 64 == 0 && p.Cin >= 128 &&
        p.Cout >= 64 && M >= 1024) {
      // static schedule: block b takes tiles b, b+G, ...; every CU runs `occ` blocks side by side, so the makespan is
      // ceil(tiles / (256 occ)) tiles per block at occ tiles per CU-time (8296 x 1024 outputs: 520 tiles of 128x128
      // on 512 resident blocks are two rounds; 464 tiles of 144x128 are one)
      // eff: relative matrix-pipe efficiency of a tile shape once resident (the 128-row tiles are bound by the texture-address
      // path: 32 KB of LDS-DMA per 512 MFMA cycles; a 256x256 tile moves half the bytes per MFMA) -- calibrated on the ViT-L linears
      struct Cand { int code, bm, bn, occ; float eff; };
      const Cand cand[6] = {{128128, 128, 128, PersistCfg<128, 128, 2, 2>::occ, 1.f}, {12896, 128, 96, PersistCfg<128, 96, 2, 2>::occ, 1.f},
                            {12864, 128, 64, PersistCfg<128, 64, 2, 2>::occ, 1.f}, {144128, 144, 128, PersistCfg<144, 128, 3, 2>::occ, 1.f},
                            {14464, 144, 64, PersistCfg<144, 64, 3, 2>::occ, 1.f},
                            {256128, 256, 128, PersistCfg<256, 128, 4, 2>::occ, PF_PERSIST_EFF_256128}};
      int code = persist;
      bool known = false;
      for (int i = 0; i < 6; ++i) known = known || cand[i].code == code;
      if (!known) {
        double best = 1e300;
        for (int i = 0; i < 6; ++i) {
          const long tiles = ((M + cand[i].bm - 1) / cand[i].bm) * ((p.Cout + cand[i].bn - 1) / cand[i].bn);
          const long rounds = (tiles + 256L * cand[i].occ - 1) / (256L * cand[i].occ);
          const double cost = (double)rounds * cand[i].occ * cand[i].bm * cand[i].bn / cand[i].eff;
          if (cost < best) { best = cost; code = cand[i].code; }
        }
      }
#define PF_PERSIST_CASE(CODE, BM_, BN_, WM_, WN_) \
      if (code == CODE) return p.relu_in ? launch_persist<BM_, BN_, WM_, WN_, true>(p, st) : launch_persist<BM_, BN_, WM_, WN_, false>(p, st);
      PF_PERSIST_CASE(128128, 128, 128, 2, 2)
      PF_PERSIST_CASE(12896, 128, 96, 2, 2)
      PF_PERSIST_CASE(12864, 128, 64, 2, 2)
      PF_PERSIST_CASE(144128, 144, 128, 3, 2)
      PF_PERSIST_CASE(14464, 144, 64, 3, 2)
      PF_PERSIST_CASE(256128, 256, 128, 4, 2)
#undef PF_PERSIST_CASE
    }
  }
  return dispatch_generic<T>(p, st, true);
}

// ---- generic implicit-GEMM path: tile choice + (f32) exact channel split -------------------------------------------
// cost ~ makespan of `blocks` tiles on 256 CUs with `occ` co-resident blocks each (blocks / slots full rounds + a tail
// round), times the per-round cost bm*bn*occ, over a small per-shape efficiency factor.
struct TileCfg { int bm, bn, occ; float eff; int id; };
static const TileCfg kCfgs[8] = {{256, 128, 1, 1.0f, 0}, {128, 128, 2, 1.0f, 1}, {128, 96, 2, 0.97f, 2},
                                 {128, 64, 3, 0.96f, 3}, {256, 32, 2, 0.75f, 4}, {256, 16, 2, 0.5f, 5}, {64, 64, 4, 0.92f, 6},
                                 {256, 256, 1, PF_F32_EFF_256, 7}};   // id 7: f32 only, eight waves (half the LDS-DMA pieces per MFMA)

template <typename T>
static double cfg_cost(const TileCfg& c, long M, int cout) {
  const long blocks = ((M + c.bm - 1) / c.bm) * ((cout + c.bn - 1) / c.bn);
  return ((double)blocks / (256.0 * c.occ) + 1.0) * c.bm * c.bn * c.occ / c.eff;
}

template <typename T>
static int best_cfg(const pf_conv_params& p, long M, int cout, double* cost_out) {
  int best = -1;
  double best_cost = 1e300;
  for (int i = 0; i < 8; ++i) {
    const TileCfg& c = kCfgs[i];
    if (c.id == 0 && (sizeof(T) != 2 || g_force_small == 1 || cout < 96)) continue;   // big kernel: bf16 only
    if (c.id == 6 && sizeof(T) != 4) continue;                                        // 64x64: f32 granularity tile
    if (c.id == 7 && (sizeof(T) != 4 || c.eff <= 0.f || cout % 256 || p.shuffle > 1)) continue;   // 256x256: f32, exact channel multiples
    double cost = cfg_cost<T>(c, M, cout);
    // f32, M below ~300k output pixels (the ViT-L linears at 8296 tokens, the 112x148 / 56x74 pyramid levels): the 64x64
    // tile measured fastest -- many small blocks beat every "fewer rounds of bigger tiles" candidate, incl. a 144-row
    // six-wave tile that quantises M = 8296 exactly (profiles/r2_f32_tune_144.log: qkv 109 vs 102 (model's choice) vs 85)
    if (c.id == 6 && M < 300000) cost *= 0.85;
    if (cost < best_cost) { best_cost = cost; best = c.id; }
  }
  if (cost_out) *cost_out = best_cost;
  return best;
}

template <typename T>
int launch_generic(const pf_conv_params& p, hipStream_t st, int cfg) {
  switch (cfg) {
    case 0: if constexpr (sizeof(T) == 2) return p.relu_in ? launch_big<true>(p, st) : launch_big<false>(p, st);
    case 1: return launch_cfg<T, 128, 128, 2, 2>(p, st);
    case 2: return launch_cfg<T, 128, 96, 2, 2>(p, st);
    case 3: return launch_cfg<T, 128, 64, 2, 2>(p, st);
    case 4: return launch_cfg<T, 256, 32, 4, 1>(p, st);
    case 6: if constexpr (sizeof(T) == 4) return launch_cfg<T, 64, 64, 2, 2>(p, st);
    case 7: if constexpr (sizeof(T) == 4) return launch_cfg<T, 256, 256, 4, 2>(p, st);
    default: return launch_cfg<T, 256, 16, 4, 1>(p, st);
  }
}

// f32 only: the matrix pipe is the ONLY resource that matters at 1/16 of the bf16 rate, so padded output channels are
// pure loss (544 = 6 x 96 - 32: 5.9 %).  Split the channel range into a body that a wide tile covers exactly and a
// remainder launched with a narrower tile (544 = 5 x 96 + 64): two launches on the same stream, zero padded MFMAs.
template <typename T>
int dispatch_generic(const pf_conv_params& p, hipStream_t st, bool allow_split) {
  const long M = (long)p.B * p.OH * p.OW * (p.batch > 1 ? p.batch : 1);   // cost model only: the planes of a batched GEMM are one stream of tiles
  // PF_IGEMM_CFG / PF_IGEMM_NOSPLIT are kernel-tuning overrides, read per call (a getenv is ~50 ns against a >= 2 us launch)
  const char* e = getenv("PF_IGEMM_CFG");
  const int force_cfg = e ? atoi(e) : -1;
  const char* e2 = getenv("PF_IGEMM_NOSPLIT");
  const int no_split = e2 ? atoi(e2) : 0;
  double cost_single;
  int best = best_cfg<T>(p, M, p.Cout, &cost_single);
  if (force_cfg >= 0 && !(force_cfg == 0 && sizeof(T) != 2) && !(force_cfg >= 6 && sizeof(T) != 4)) return launch_generic<T>(p, st, force_cfg);
  // batched planes (the transform-domain GEMMs of a Winograd layer): K is one layer's Cin (17 .. 32 chunks), so the per-tile
  // prologue / epilogue weighs as much as the K loop -- the 128x64 tile with three blocks per CU overlaps them best on every
  // shape measured (profiles/r2c_wino_tune.log: 118 vs 111 TF/s (cost model's split) at 544->544, 130 vs 121 at 768->768)
  if (p.batch > 1) return launch_generic<T>(p, st, p.Cout <= 32 ? 4 : 3);
  if constexpr (sizeof(T) == 4) {
    if (allow_split && !no_split && p.shuffle <= 1 && p.Cout > 128) {
      int split_at = 0, split_cfg = -1;
      double split_cost = cost_single * 0.985;        // a second launch must buy at least 1.5 %
      for (int i = 1; i <= 7; ++i) {                  // body tiles 128x128 / 128x96 / 128x64 / 256x256
        if (i > 3 && i < 7) continue;
        const TileCfg& c = kCfgs[i];
        if (i == 7 && c.eff <= 0.f) continue;
        const int body = (p.Cout / c.bn) * c.bn, rem = p.Cout - body;
        if (body == 0 || rem == 0) continue;
        double rem_cost;
        best_cfg<T>(p, M, rem, &rem_cost);
        const double cost = cfg_cost<T>(c, M, body) + rem_cost;
        if (cost < split_cost) { split_cost = cost; split_at = body; split_cfg = c.id; }
      }
      if (split_at > 0) {
        pf_conv_params a = p, b = p;
        a.Cout = split_at;
        b.Cout = p.Cout - split_at;
        b.w = reinterpret_cast<const char*>(p.w) + (size_t)split_at * p.Kpad * sizeof(T);
        b.w_rows = p.w_rows - split_at;
        if (p.bias) b.bias = p.bias + split_at;
        if (p.scale) b.scale = p.scale + split_at;
        if (p.res) b.res = reinterpret_cast<const char*>(p.res) + (size_t)split_at * sizeof(T);
        if (p.res2) b.res2 = reinterpret_cast<const char*>(p.res2) + (size_t)split_at * sizeof(T);
        b.y = reinterpret_cast<char*>(p.y) + (size_t)split_at * sizeof(float);   // f32 path stores float
        int rc = launch_generic<T>(a, st, split_cfg);
        if (rc) return rc;
        return dispatch_generic<T>(b, st, false);
      }
    }
  }
  return launch_generic<T>(p, st, best);
}

int validate(const pf_conv_params* p) {
  const int vec = p->dtype == PF_DTYPE_BF16 ? 8 : 4;
  const int bk = 8 * vec;
  const char* e = nullptr;
  if (!p->x || !p->w || !p->y) e = "null tensor";
  else if (p->dtype != PF_DTYPE_F32 && p->dtype != PF_DTYPE_BF16) e = "bad dtype";
  else if (p->Cin <= 0 || p->Cin % vec) e = "Cin must be a positive multiple of the 16-byte vector";
  else if (p->x_ld % vec || p->x_ld < p->Cin) e = "x_ld must be a multiple of the vector and >= Cin";
  else if (p->Cout <= 0 || p->Cout % 4) e = "Cout must be a positive multiple of 4";
  else if (p->y_ld % 4 || (p->res && p->res_ld % 4) || (p->res2 && p->res2_ld % 4)) e = "output/residual ld must be multiples of 4";
  else if (p->Kpad % bk) e = "Kpad must be a multiple of the 128-byte chunk";
  else if ((long)p->KH * p->KW * (p->Cin / vec) > (long)(p->Kpad / vec)) e = "Kpad smaller than KH*KW*Cin";
  else if (p->w_rows < p->Cout) e = "w_rows < Cout";
  else if (p->shuffle > 1 && (p->Cout % (p->shuffle * p->shuffle) || (p->Cout / (p->shuffle * p->shuffle)) % 4)) e = "shuffle: Cout/(s*s) must be a multiple of 4";
  else if (p->shuffle > 1 && (p->KH != 1 || p->KW != 1 || p->res || p->res2)) e = "shuffle only for 1x1 without residual";
  else if (p->KH * p->KW > 16) e = "at most 16 filter taps";
  else if ((long)p->B * p->OH * p->OW <= 0) e = "empty output";
  else if ((long)p->B * p->OH * p->OW >= (1L << 31)) e = "too many output pixels";
  else if (p->korder != 0 && (p->korder != 1 || p->dtype != PF_DTYPE_F32 || p->Cin % 32 || p->shuffle > 1)) e = "korder 1 needs f32, Cin % 32 == 0, no shuffle";
  else if (p->batch > 1 && (p->dtype != PF_DTYPE_F32 || p->shuffle > 1 || p->res || p->res2 || p->bias || p->scale || p->batch > 65535))
    e = "batch > 1: float32 GEMM planes without bias / scale / residual only";
  if (e) { snprintf(g_err, sizeof(g_err), "pf_conv: %s", e); return PF_ERR_ARG; }
  return PF_OK;
}

}  // namespace

extern "C" const char* pf_last_error(void) { return g_err; }
extern "C" int pf_version(void) { return 1; }

extern "C" int pf_conv(const pf_conv_params* p, void* stream) {
  if (!p) return PF_ERR_ARG;
  int rc = validate(p);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  rc = p->dtype == PF_DTYPE_BF16 ? dispatch<bf16_t>(*p, st) : dispatch<float>(*p, st);
  if (rc) snprintf(g_err, sizeof(g_err), "pf_conv: launch failed: %s", hipGetErrorString(g_launch_status));
  return rc;
}

extern "C" int pf_conv_timed(const pf_conv_params* p, int iters, float* ms, void* stream) {
  if (!p || !ms || iters <= 0) return PF_ERR_ARG;
  int rc = validate(p);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  rc = pf_conv(p, stream);  // warm-up
  hipEventRecord(e0, st);
  for (int i = 0; i < iters && rc == PF_OK; ++i) rc = pf_conv(p, stream);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  *ms = t / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}
