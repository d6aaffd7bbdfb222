// Split-precision linear layer: float32-grade GEMM on the bf16 matrix cores.  Default for the ViT block linears of the float32 mode since its
// error against float64 was measured equal to the f32 MFMA kernel's (tests/op_checks.py gemm_split3); PF_LINEAR_SPLIT3=0 turns it off.
//
// The f32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 rate, and the ViT-L linear layers (dinov2/layers/attention.py:51,60,
// mlp.py:35-41: qkv / proj / fc1 / fc2, 29 % of the float32 image pass) already sit at 0.69-0.78 of that peak.  Here every float32
// operand is carried as THREE bf16 planes  x = x_h + x_m + x_l  (round-to-nearest splits: 8 + 8 + 8 significant bits, exact for
// 1e-30 < |x| < 3.39e38, the bf16 maximum) and the product is evaluated as the six leading partial products
//     x.w ~= x_l.w_h + x_h.w_l + x_m.w_m + x_m.w_h + x_h.w_m + x_h.w_h          (dropped: x_m.w_l, x_l.w_m, x_l.w_l <= 2^-24 |x||w|)
// on v_mfma_f32_16x16x32_bf16 with float32 accumulators, smallest terms first.  Each bf16 x bf16 product is exact in float32, so the
// result differs from the float32 FMA chain by a few 2^-24 relative to sum |x||w| -- float32 rounding class, not bf16 (measured against
// float64 in tests/op_checks.py gemm_split3).  Six MFMAs at 16x the f32 rate = 2.67x the float32 MFMA peak.
//
//   D[n][m] = sum_k W[n][k] X[m][k]      n = output channel, m = token; MFMA A = weight rows, B = token rows (a lane's four accumulator
//                                         registers = four consecutive channels of one token, as in igemm.hip)
// Operands: X planes [3][M][x_ld] bf16 (plane stride x_bstride elements), W planes [3][w_rows][Kpad] (w_bstride), K = Cin % 32 == 0.
// Output: float32 [M][y_ld] (out_f32 = 1) or, for a following split GEMM (fc1 -> fc2), three bf16 planes [3][M][y_ld] (y_bstride).
// Epilogue in float32 exactly like pf_conv: (act(v + bias) * scale) + res + res2.
//
// Tile 128 x 128 (or 64 x 128), K chunks of 32 (64-byte rows), eight (four) waves of 32 x 64; both operands' three planes are staged by
// LDS-DMA into a 2-deep ring (48 KiB per stage).  64-byte rows: 16-byte slot g of row r sits at slot g ^ ((r >> 1) & 3), which makes the
// 16-lane groups of ds_read_b128 hit 16 distinct 16-byte positions of the 256-byte bank window (exhaustive check: tools/lds_swizzle_check.py);
// the DMA writes lane-linearly, so the swizzle is applied on the source side.
#include <atomic>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned int s3_zero_page[64];

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// barrier that does not drain the DMA queue (__syncthreads may be lowered with s_waitcnt vmcnt(0))
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// barrier without waiting for this wave's own LDS reads (ping-pong schedule: the reads issued just before it may stay in flight)
__device__ __forceinline__ void plain_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// NP = bf16 planes per operand and stage.  PLAIN = false: the split-precision GEMM (NP = 3 planes h / m / l, six cross terms, float32 residuals,
// float32 or three-plane output).  PLAIN = true: an ordinary bf16 GEMM through the same pipeline -- the "planes" are NP consecutive 32-deep K
// sub-chunks of ONE bf16 matrix (plane stride = 32 elements, the stage advances K by 32 NP), the terms are the NP diagonal products, residuals
// and output are bf16 (float32 output when out_f32).
template <int BM, int BN, int WM, int WN, int NS, bool PP = false, int NP = 3, bool PLAIN = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_split3_kernel(const pf_conv_params p) {
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int ROWS = NP * (BM + BN), PIECES = ROWS / 16, PPW = PIECES / NW;      // 16 rows of 64 B per 1-KiB DMA piece
  constexpr int STAGE = ROWS * 64;
  static_assert(PIECES % NW == 0 && BM % 16 == 0 && BN % 16 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = p.B * p.OH * p.OW;
  const int nt = (p.Cout + BN - 1) / BN;
  const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  // groups of 8 token tiles x all channel tiles (column by column), like igemm.hip tile_of: the blocks an XCD runs at once share panels in L2
  int tile_m, tile_n;
  {
    const int mt = (M + BM - 1) / BM, per_group = 8 * nt;
    const int group = bid / per_group, first = group * 8;
    const int gsz = min(mt - first, 8), in_g = bid - group * per_group;
    tile_n = in_g / gsz;
    tile_m = first + (in_g - tile_n * gsz);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- loader: wave w moves pieces w*PPW .. +PPW-1; lane L -> row 16 q + (L >> 2), physical slot L & 3 = logical slot ^ ((row >> 1) & 3)
  // batch > 1 (the transform points of a Winograd layer, csrc/winograd.hip run_split3): point blockIdx.y reads the [M][x_ld] / [w_rows][Kpad]
  // block number blockIdx.y of every plane and writes the float32 block [M][y_ld] number blockIdx.y
  const long bz = blockIdx.y;
  const bf16_t* __restrict__ xg = reinterpret_cast<const bf16_t*>(p.x) + bz * (long)M * p.x_ld;
  const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.w) + bz * (long)p.w_rows * p.Kpad;
  const long y_base = bz * (long)M * p.y_ld;
  const char* zero = reinterpret_cast<const char*>(s3_zero_page);
  const char* cur[PPW];
  int inc[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = (wave * PPW + i) * 16 + (lane >> 2);        // row of the stage: [X plane 0..2 | W plane 0..2]
    const int j = (lane & 3) ^ ((row >> 1) & 3);
    const char* src = zero;
    if (row < NP * BM) {
      const int pl = row / BM, m = m0 + (row - pl * BM);
      if (m < M) src = reinterpret_cast<const char*>(xg + (size_t)pl * p.x_bstride + (size_t)m * p.x_ld + j * 8);
    } else {
      const int rw = row - NP * BM, pl = rw / BN, n = n0 + (rw - pl * BN);
      if (n < p.w_rows) src = reinterpret_cast<const char*>(wg + (size_t)pl * p.w_bstride + (size_t)n * p.Kpad + j * 8);
    }
    cur[i] = src;
    inc[i] = src == zero ? 0 : (PLAIN ? 64 * NP : 64);
  }
  const unsigned smem_base = lds_addr(smem);
#ifdef PF_S3_DBG           // timing decomposition (results wrong by construction): env PF_S3_DBG bit 0 = no DMA after the first ring fill,
  const int dbg = p.pad;   // bit 1 = no MFMA, bit 2 = no fragment reads after the first
#define S3_DBG(bit) (dbg & (bit))
#else
#define S3_DBG(bit) 0
#endif
  int issued = 0;
  auto issue = [&](int stage) {
    if (S3_DBG(1) && issued >= NS) return;
    ++issued;
    const unsigned dst = smem_base + stage * STAGE + wave * (PPW * 1024);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      glds16(cur[i], dst + i * 1024);
      cur[i] += inc[i];
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int slot = (fg ^ ((fr >> 1) & 3)) << 4;                    // fragment rows are multiples of 16 apart: the swizzle term is per lane
  const int x_off = (wm * WTM + fr) * 64 + slot;                   // + plane * BM * 64 + fm * 1024
  const int w_off = NP * BM * 64 + (wn * WTN + fr) * 64 + slot;     // + plane * BN * 64 + fn * 1024
  // per-channel epilogue constants before the K loop (their latency hides behind it)
  float4 bias_r[FN], scale_r[FN];
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

  // Ring of NS = 3 stage slots, fragments double-buffered in registers: while chunk kc is multiplied out of registers, the fragments of chunk
  // kc+1 are read from LDS (the matrix pipe never waits for ds_read), chunk kc+2 is in flight and chunk kc+3 is issued into the slot chunk kc
  // just left.  ONE barrier per chunk: it publishes stage kc+1 and retires the reads of stage kc.
  const int nk = PLAIN ? p.Cin / (32 * NP) : p.Cin / 32;
  struct Frags { uint4 w[NP][FN], x[NP][FM]; };
  auto read_frags = [&](Frags& f, int kc) {
    if (S3_DBG(4) && kc > 1) return;
    const char* S = smem + (kc % NS) * STAGE;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) f.w[pl][fn] = *reinterpret_cast<const uint4*>(S + w_off + pl * (BN * 64) + fn * 1024);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) f.x[pl][fm] = *reinterpret_cast<const uint4*>(S + x_off + pl * (BM * 64) + fm * 1024);
    }
  };
  // six partial products, smallest first; within a term the FN*FM accumulators are independent chains.  (Skipping the channel fragments
  // beyond Cout -- 544 = 4 x 128 + 32 -- behind a wave-uniform test was measured: 3-7 % SLOWER on every shape, the test breaks the MFMA
  // schedule; profiles/r3_three_step_split.log vs r3aj)
  auto multiply = [&](const Frags& f) {
    if (S3_DBG(2)) return;
#define S3_TERM(PW, PX)                                                                                                      \
  _Pragma("unroll") for (int fn = 0; fn < FN; ++fn) _Pragma("unroll") for (int fm = 0; fm < FM; ++fm)                       \
      acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.w[PW][fn]), __builtin_bit_cast(bf16x8, f.x[PX][fm]), \
                                                            acc[fn][fm], 0, 0, 0);
    if constexpr (PLAIN) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) { S3_TERM(pl, pl) }
    } else {
      S3_TERM(0, 2) S3_TERM(2, 0) S3_TERM(1, 1) S3_TERM(0, 1) S3_TERM(1, 0) S3_TERM(0, 0)
    }
#undef S3_TERM
  };
  auto step = [&](const Frags& cur, Frags& nxt, int kc) {
    if (kc + 1 < nk) {
      if (kc + 2 < nk) vm_wait<PPW>();                  // this wave's pieces of stage kc+1 have landed (stage kc+2 may still fly) ...
      else vm_wait<0>();
      lds_barrier();                                    // ... and every wave's; all waves hold chunk kc in registers: its slot is free
      if (kc + 3 < nk) issue(kc % NS);
      read_frags(nxt, kc + 1);
    }
    multiply(cur);
  };
  if constexpr (NS == 3 && PP) {
    // PING-PONG: the two waves of a SIMD (wave w and w + NW/2) run the same loop ONE PHASE apart.  A chunk is two phases separated by
    // barriers -- M(kc): issue the DMA pieces of chunk kc+3, read the fragments of chunk kc+1, wait for this wave's pieces of chunk kc+2;
    // C(kc): the 48 MFMAs of chunk kc out of registers -- and group B enters the loop one barrier later than group A, so that in every phase
    // one wave of each SIMD feeds the matrix pipe while the other does its loads.  Hazards (phases numbered globally, A: M(kc) = 2kc,
    // B: M(kc) = 2kc+1): chunk kc+1 is read in M(kc); its pieces were waited for at the end of M(kc-1) by both groups (phases 2kc-2, 2kc-1);
    // the slot of chunk kc is overwritten from phase 2kc on, after its last read in phase 2kc-1 (lgkmcnt(0) before that phase's barrier).
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (i < nk) issue(i);
    if (nk > 2) vm_wait<PPW>();                         // chunks 0 and 1 landed
    else vm_wait<0>();
    lds_barrier();
    Frags fa, fb;
    read_frags(fa, 0);
    lds_barrier();
    const bool grp_b = wave >= NW / 2;
    if (grp_b) plain_barrier();                         // one phase behind
    auto pp_chunk = [&](const Frags& cur, Frags& nxt, int kc) {
      if (kc + 3 < nk) issue(kc % NS);
      if (kc + 1 < nk) read_frags(nxt, kc + 1);
      if (kc + 3 < nk) vm_wait<PPW>();
      else if (kc + 2 < nk) vm_wait<0>();
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      multiply(cur);
      __builtin_amdgcn_s_setprio(0);
      plain_barrier();
    };
    for (int kc = 0; kc < nk; kc += 2) {
      pp_chunk(fa, fb, kc);
      if (kc + 1 < nk) pp_chunk(fb, fa, kc + 1);
    }
    if (!grp_b) plain_barrier();
  } else if constexpr (NS == 3) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (i < nk) issue(i);
    if (nk > 2) vm_wait<2 * PPW>();
    else if (nk > 1) vm_wait<PPW>();
    else vm_wait<0>();
    lds_barrier();
    Frags fa, fb;
    read_frags(fa, 0);
    for (int kc = 0; kc < nk; kc += 2) {
      step(fa, fb, kc);
      if (kc + 1 < nk) step(fb, fa, kc + 1);
    }
  } else {
    // two-slot ring, fragments read at the top of each chunk (half the registers: two blocks per CU for the 64-token tile)
    static_assert(NS == 2, "ring depth");
    issue(0);
    for (int kc = 0; kc < nk; ++kc) {
      vm_wait<0>();
      lds_barrier();                                    // stage kc landed for every wave; all waves are done reading stage kc-1
      if (kc + 1 < nk) issue((kc + 1) & 1);
      Frags f;
      read_frags(f, kc);
      multiply(f);
    }
  }

  // ---- epilogue: bias -> act -> scale -> residual(s) -> float32 store or three-plane split store ----
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int m = m0 + wm * WTM + fm * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = n0 + wn * WTN + fn * 16 + fg * 4;
      if (n >= p.Cout) continue;
      float v[4] = {acc[fn][fm][0] + bias_r[fn].x, acc[fn][fm][1] + bias_r[fn].y, acc[fn][fm][2] + bias_r[fn].z, acc[fn][fm][3] + bias_r[fn].w};
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
      if constexpr (PLAIN) {
        if (p.res) {
          float a[4];
          load4(reinterpret_cast<const bf16_t*>(p.res) + (long)m * p.res_ld + n, a);
          v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
        }
        if (p.res2) {
          float a[4];
          load4(reinterpret_cast<const bf16_t*>(p.res2) + (long)m * p.res2_ld + n, a);
          v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
        }
        if (p.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (long)m * p.y_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
        else store4(reinterpret_cast<bf16_t*>(p.y) + (long)m * p.y_ld + n, v[0], v[1], v[2], v[3]);
      } else {
      if (p.res) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (long)m * p.res_ld + n);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      }
      if (p.res2) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res2) + (long)m * p.res2_ld + n);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      }
      if (p.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + y_base + (long)m * p.y_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
      else store_split3(reinterpret_cast<bf16_t*>(p.y) + (long)m * p.y_ld + n, p.y_bstride, v);
      }
    }
  }
}

thread_local char g_err[200] = {0};

template <int BM, int BN, int WM, int WN, int NS, bool PP = false, int NP = 3, bool PLAIN = false>
int launch(const pf_conv_params& p, hipStream_t st) {
  constexpr int smem = NS * NP * (BM + BN) * 64;
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  auto kern = gemm_split3_kernel<BM, BN, WM, WN, NS, PP, NP, PLAIN>;
  if (!(done.load(std::memory_order_acquire) & bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    done.fetch_or(bit, std::memory_order_release);
  }
  const long M = (long)p.B * p.OH * p.OW;
  const long tiles = ((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(64 * WM * WN), smem, st, p);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

}  // namespace

extern "C" int pf_gemm_split3(const pf_conv_params* p, void* stream) {
  const char* e = nullptr;
  if (!p || !p->x || !p->w || !p->y) return PF_ERR_ARG;
#ifdef PF_S3_DBG
  pf_conv_params pd = *p;
  if (const char* s = getenv("PF_S3_DBG")) pd.pad = atoi(s);
  p = &pd;
  if (p->KH != 1 || p->KW != 1 || p->stride != 1 || p->shuffle > 1) e = "1x1 / linear layers only";
#else
  if (p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad != 0 || p->shuffle > 1) e = "1x1 / linear layers only";
#endif
  else if (p->Cin <= 0 || p->Cin % 32 || p->Kpad < p->Cin || p->Kpad % 32 || p->x_ld % 8 || p->x_ld < p->Cin) e = "K must be a multiple of 32, x_ld of 8";
  else if (p->Cout <= 0 || p->Cout % 4 || p->y_ld % 4 || p->w_rows < p->Cout) e = "Cout / y_ld must be multiples of 4";
  else if ((p->res && p->res_ld % 4) || (p->res2 && p->res2_ld % 4)) e = "residual ld must be a multiple of 4";
  else if ((long)p->B * p->OH * p->OW <= 0 || (long)p->B * p->OH * p->OW >= (1L << 31)) e = "bad token count";
  else if (p->x_bstride <= 0 || p->w_bstride <= 0 || (!p->out_f32 && p->y_bstride <= 0)) e = "plane strides missing";
  else if (p->batch > 1 && (!p->out_f32 || p->bias || p->scale || p->res || p->res2 || p->batch > 65535)) e = "batched planes: float32 output, no epilogue";
  if (e) return PF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // tile: 128 x 128 (eight waves, one block per CU, ping-pong) unless the token x channel grid does not fill the chip once; PF_S3_TILE_NOW forces
  int force = 0;
  if (const char* s = getenv("PF_S3_TILE_NOW")) force = atoi(s);      // (tests: read per call)
  const long M = (long)p->B * p->OH * p->OW;
  const long t128 = ((M + 127) / 128) * ((p->Cout + 127) / 128);
  const bool small = force ? force == 64 : t128 < 256;       // (ping-pong 128 x 128 wins from one full round of tiles on: profiles/r3_split3_pingpong.log)
    if (small) return launch<64, 128, 2, 2, 2>(*p, st);
  const char* pp = getenv("PF_S3_PP");                    // (A/B switch; read per call)
  return (pp && pp[0] == '0') ? launch<128, 128, 4, 2, 3, false>(*p, st) : launch<128, 128, 4, 2, 3, true>(*p, st);
}

// plain bf16 linear layer through the ping-pong pipeline (see the PLAIN template parameter): x [M][x_ld] bf16, w [w_rows][Kpad] bf16
// (packing.pack_conv), bias / scale float32, res / res2 / y bf16 (y float32 when out_f32); Cin % 64 == 0.  Tile 256 x 128, eight waves of 64 x 64.
extern "C" int pf_gemm_bf16_pp(const pf_conv_params* p, void* stream) {
  if (!p || !p->x || !p->w || !p->y) return PF_ERR_ARG;
  if (p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad != 0 || p->shuffle > 1 || p->dtype != PF_DTYPE_BF16) return PF_ERR_ARG;
  if (p->Cin <= 0 || p->Cin % 64 || p->Kpad < p->Cin || p->x_ld % 8 || p->x_ld < p->Cin || p->Kpad % 8) return PF_ERR_ARG;
  if (p->Cout <= 0 || p->Cout % 4 || p->y_ld % 4 || p->w_rows < p->Cout || (p->res && p->res_ld % 4) || (p->res2 && p->res2_ld % 4)) return PF_ERR_ARG;
  if ((long)p->B * p->OH * p->OW <= 0 || (long)p->B * p->OH * p->OW >= (1L << 31)) return PF_ERR_ARG;
  pf_conv_params pd = *p;
  pd.x_bstride = 32;                     // "planes" = consecutive 32-deep K sub-chunks of the one bf16 matrix
  pd.w_bstride = 32;
  return launch<256, 128, 4, 2, 3, true, 2, true>(pd, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pf_gemm_split3_timed(const pf_conv_params* p, int iters, float* ms, void* stream) {
  if (!ms || iters <= 0) return PF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = pf_gemm_split3(p, stream);
  if (rc != PF_OK) return rc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < iters && rc == PF_OK; ++i) rc = pf_gemm_split3(p, stream);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  *ms = t / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}

// float32 [rows][x_ld] -> three bf16 planes [3][rows][y_ld] (round-to-nearest splits); the stand-alone form of the split that the producers
// (LayerNorm, attention, the split GEMM's own epilogue) otherwise fuse into their stores
__global__ void split3_kernel(const float* __restrict__ x, int x_ld, bf16_t* __restrict__ y, int y_ld, long plane, long rows, int cols) {
  const int cv = cols >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cv; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cv;
    const int c = (int)(i - r * cv) * 4;
    const float4 a = *reinterpret_cast<const float4*>(x + r * x_ld + c);
    const float v[4] = {a.x, a.y, a.z, a.w};
    store_split3(y + r * y_ld + c, plane, v);
  }
}

extern "C" int pf_split3(const float* x, int x_ld, void* y, int y_ld, long plane, long rows, int cols, void* stream) {
  if (!x || !y || cols % 4 || x_ld % 4 || y_ld % 4 || rows <= 0 || plane <= 0) return PF_ERR_ARG;
  long n = rows * (cols / 4);
  long g = (n + 255) / 256;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_ld,
                     static_cast<bf16_t*>(y), y_ld, plane, rows, cols);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}
