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
// CHUNK-MAJOR operands (round 4; p.korder bit 1 = X, bit 2 = W): a plane is stored as [K/32][rows][32] -- every 32-deep K chunk of all rows is one
// contiguous slab, so the 16 rows x 64 B of a DMA piece are ONE contiguous KiB (eight full 128-byte lines).  In the row-major layout a piece
// touches 16 lines and uses half of each; the other half is fetched again one chunk later, after 48 KiB of other traffic went through the 32-KiB
// vector L1: the L2 -> L1 traffic of the kernel is twice its operand bytes, and that stream -- not the matrix pipe -- bounds it
// (profiles/r4_persist_decomp.md: DMA stream alone 9.3 ms, MFMA stream alone 7.8 ms, full kernel 12.2 ms on the dominant launch).
// Output: float32 [M][y_ld] (out_f32 = 1) or, for a following split GEMM (fc1 -> fc2), three bf16 planes [3][M][y_ld] (y_bstride).
// Epilogue in float32 exactly like pf_conv: (act(v + bias) * scale) + res + res2.
//
// Tile 128 x 128 (or 64 x 128), K chunks of 32 (64-byte rows), eight (four) waves of 32 x 64; both operands' three planes are staged by
// LDS-DMA into a 2-deep ring (48 KiB per stage).  64-byte rows: 16-byte slot g of row r sits at slot g ^ ((r >> 1) & 3), which makes the
// 16-lane groups of ds_read_b128 hit 16 distinct 16-byte positions of the 256-byte bank window (exhaustive check: tools/lds_swizzle_check.py);
// the DMA writes lane-linearly, so the swizzle is applied on the source side.
#include <atomic>
#include <type_traits>
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
    int step = PLAIN ? 64 * NP : 64;
    if (row < NP * BM) {
      const int pl = row / BM, m = m0 + (row - pl * BM);
      if (m < M) src = reinterpret_cast<const char*>(xg + (size_t)pl * p.x_bstride + (size_t)m * ((p.korder & 2) ? 32 : p.x_ld) + j * 8);
      if (p.korder & 2) step = M * 64;                  // chunk-major: the next K chunk of this row is one [M][32] slab further
    } else {
      const int rw = row - NP * BM, pl = rw / BN, n = n0 + (rw - pl * BN);
      if (n < p.w_rows) src = reinterpret_cast<const char*>(wg + (size_t)pl * p.w_bstride + (size_t)n * ((p.korder & 4) ? 32 : p.Kpad) + j * 8);
      if (p.korder & 4) step = p.w_rows * 64;
    }
    cur[i] = src;
    inc[i] = src == zero ? 0 : step;
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
      else store_split3(reinterpret_cast<bf16_t*>(p.y) + split3_at(m, n, p.y_ld, (p.korder & 8) ? M : 0), p.y_bstride, v);
      }
    }
  }
}

// ---- PERSISTENT ping-pong form (round 4) -----------------------------------------------------------------------------------------------------
// One resident block per CU walks its share of the 128 x 128 output tiles; the K chunks of consecutive tiles form ONE stream through the
// three-slot ring, so the first three chunks of tile t+1 are in flight under the last MFMAs of tile t, and the epilogue of tile t (stores only;
// the accumulators are re-zeroed) sits at the head of the load phase M of tile t+1's first chunk, i.e. beside the partner group's MFMAs.
// The one-tile-per-block kernel above pays launch + ring fill + drain + store tail per tile with nothing else resident on the CU (144 KiB of
// LDS): ~10 us of a 22-25 us tile at K = 544 (profiles/r3_split3_gemm_decomp.log: 75 of 369 us with no MFMA, DMA or fragment read at all).
// Tile order: block b runs on XCD b % 8 (observed; locality only).  XCD x owns the contiguous tile range [total x / 8, total (x+1) / 8); in
// iteration i its local block j takes tile lo + i nb + j, so the ~32 tiles an XCD works on at once are consecutive in the order
//   plane z  >  groups of gm token tiles  >  channel tile  >  token tile within the group           (gm = 32 / nt for nt <= 6, else 8)
// = a gm x (32 / gm) patch whose X and W panels are fetched into that XCD's L2 once and shared.
// Loader: every DMA piece (16 rows of one plane) has a wave-uniform 64-bit base in SGPRs (tile origin + plane, advanced by 64 B per chunk with
// scalar adds) and a per-lane 32-bit offset (row x ld + swizzled slot) that changes only with the tile's row clamp: rows beyond M / w_rows
// are clamped to the last valid row instead of reading a zero page -- their products land in accumulator rows / columns that are never stored.
// BARE: plain float32 store (the batched transform-domain GEMM of a Winograd layer); otherwise the full pf_conv epilogue.
__device__ __forceinline__ void glds16s(unsigned voff, unsigned long long sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// (Measured and NOT kept, round 4: issuing the last 2 .. 6 of a wave's six DMA pieces per chunk from the MFMA phase, spread between the product
// terms, instead of from the load phase -- 0.97-1.03x on the batched Winograd GEMMs, 0.75-1.18x on the ViT linears, the image pass slower
// (profiles/r4_p2_sweep.md).  A piece stalls its wave ~95 cycles wherever it is issued; the load phase is what the 192 x 192 kernel below shortens.)
template <bool BARE>
__global__ __launch_bounds__(512) void gemm_split3_persist_kernel(const pf_conv_params p, int mt, int nt, int gm, int total) {
  constexpr int BM = 128, BN = 128, WM = 4, WN = 2, NP = 3, NS = 3;
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int ROWS = NP * (BM + BN), PIECES = ROWS / 16, PPW = PIECES / NW;
  constexpr int STAGE = ROWS * 64;
  static_assert(PPW * 16 * WM == NP * BM && WM * 2 == NW && BM == BN, "waves 0..3 stage the X planes, waves 4..7 the W planes");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = p.B * p.OH * p.OW;
  const int nk = p.Cin / 32;

  // ---- this block's tiles
  const int G = (int)gridDim.x, xcd = (int)blockIdx.x & 7, nb = (G - xcd + 7) >> 3;
  const int lo = (int)((long)total * xcd / 8), hi = (int)((long)total * (xcd + 1) / 8);
  const int l_first = lo + ((int)blockIdx.x >> 3);
  if (l_first >= hi) return;
  const int my_tiles = (hi - l_first + nb - 1) / nb;
  const int chunks = my_tiles * nk;
  const int per_plane = mt * nt, per_group = gm * nt;
  auto decode = [&](int l, int& z, int& m0, int& n0) __attribute__((always_inline)) {
    z = l / per_plane;
    const int r = l - z * per_plane;
    if (gm == 0) {                                        // channel tile fastest (tile_group: nt does not divide an XCD's 32 blocks)
      const int t = r / nt;
      m0 = t * BM;
      n0 = (r - t * nt) * BN;
      return;
    }
    const int group = r / per_group, in_g = r - group * per_group, first = group * gm;
    const int gsz = min(mt - first, gm);
    const int tn = in_g / gsz;
    m0 = (first + in_g - tn * gsz) * BM;
    n0 = tn * BN;
  };

  // ---- loader: wave w moves pieces w*PPW .. +PPW-1 of a stage [X h | X m | X l | W h | W m | W l]; lane L -> row 16 q + (L >> 2),
  // physical slot L & 3 = logical slot ^ ((row >> 1) & 3)
  const bool is_x = wave < WM;
  const bool kmaj = (p.korder & (is_x ? 2 : 4)) != 0;                         // chunk-major operand: [K/32][rows][32]
  const int lim = is_x ? M : p.w_rows;
  const int ld_b = kmaj ? 64 : (is_x ? p.x_ld : p.Kpad) * 2;                  // operand row pitch in bytes
  const unsigned adv_b = kmaj ? (unsigned)lim * 64u : 64u;                    // from one K chunk to the next
  const unsigned long long op_base = is_x ? (unsigned long long)p.x : (unsigned long long)p.w;
  const unsigned long long pl_b = (unsigned long long)(is_x ? p.x_bstride : p.w_bstride) * 2;   // h / m / l plane pitch in bytes
  const unsigned long long z_b = (unsigned long long)lim * ((is_x ? p.x_ld : p.Kpad) * 2);      // transform-point pitch (batched) in bytes
  unsigned long long sbase[PPW];
  unsigned voff[PPW];
  auto setup_loader = [&](int l) __attribute__((always_inline)) {
    int z, m0, n0;
    decode(l, z, m0, n0);
    const int origin = is_x ? m0 : n0;
    const unsigned long long tb = op_base + (unsigned long long)z * z_b + (unsigned long long)origin * ld_b;
    const int last = lim - 1 - origin;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int row = ((wave & (WM - 1)) * PPW + i) * 16 + (lane >> 2);          // row within this operand's three planes
      const int j = (lane & 3) ^ ((row >> 1) & 3);                               // (BM is a multiple of 4: same swizzle as the stage row)
      const int pl = ((wave & (WM - 1)) * PPW + i) / (BM / 16), r = row - pl * BM;
      sbase[i] = tb + pl * pl_b;
      voff[i] = (unsigned)(min(r, last) * ld_b + j * 16);
    }
  };
  const unsigned smem_base = lds_addr(smem);
  // the loader's cursor is advanced at the HEAD of an issue (ring slot, chunk within the tile, tile switch), always in a load phase
  int l_load = l_first, l_kc = -1, s_issue = NS - 1;
  unsigned dst_cur = 0;
  setup_loader(l_load);
#ifdef PF_S3_DBG           // timing decomposition (results wrong by construction): p.pad bit 0 = no DMA after the first ring fill, bit 1 = no MFMA,
  const int dbg = p.pad;   // bit 2 = no fragment reads after the first, bit 3 = no epilogue stores
  int dbg_issued = 0, dbg_reads = 0;
#define S3P_DBG(bit) (dbg & (bit))
#else
#define S3P_DBG(bit) 0
#endif
  auto begin_issue = [&]() __attribute__((always_inline)) {
    s_issue = s_issue == NS - 1 ? 0 : s_issue + 1;
    if (++l_kc == nk) {                                  // the stream moves on to this block's next tile
      l_kc = 0;
      l_load += nb;
      if (l_load < hi) setup_loader(l_load);
    }
    // (readfirstlane: the ring position is wave-uniform by construction, but hipcc's divergence analysis loses that through the tile walk)
    dst_cur = __builtin_amdgcn_readfirstlane(smem_base + s_issue * STAGE + wave * (PPW * 1024));
  };
  auto piece = [&](int i) __attribute__((always_inline)) {
#ifdef PF_S3_DBG
    if (S3P_DBG(1) && dbg_issued >= NS * PPW) return;
    ++dbg_issued;
#endif
    glds16s(voff[i], sbase[i], dst_cur + i * 1024);
    sbase[i] += adv_b;
  };
  auto issue = [&]() __attribute__((always_inline)) {    // a whole chunk from the load phase
    begin_issue();
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece(i);
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int slot = (fg ^ ((fr >> 1) & 3)) << 4;
  const int x_off = (wm * WTM + fr) * 64 + slot;
  const int w_off = NP * BM * 64 + (wn * WTN + fr) * 64 + slot;
  struct Frags { uint4 w[NP][FN], x[NP][FM]; };
  int s_read = 0;
  auto read_frags = [&](Frags& f) __attribute__((always_inline)) {
#ifdef PF_S3_DBG
    if (S3P_DBG(4) && dbg_reads++ >= 2) return;
#endif
    const char* S = smem + s_read * STAGE;
    s_read = s_read == NS - 1 ? 0 : s_read + 1;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) f.w[pl][fn] = *reinterpret_cast<const uint4*>(S + w_off + pl * (BN * 64) + fn * 1024);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) f.x[pl][fm] = *reinterpret_cast<const uint4*>(S + x_off + pl * (BM * 64) + fm * 1024);
    }
  };
#define S3_TERM(PW, PX)                                                                                                      \
  _Pragma("unroll") for (int fn = 0; fn < FN; ++fn) _Pragma("unroll") for (int fm = 0; fm < FM; ++fm)                       \
      acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.w[PW][fn]), __builtin_bit_cast(bf16x8, f.x[PX][fm]), \
                                                            acc[fn][fm], 0, 0, 0);
  auto multiply = [&](const Frags& f) __attribute__((always_inline)) {
    if (S3P_DBG(2)) return;
    S3_TERM(0, 2) S3_TERM(2, 0) S3_TERM(1, 1) S3_TERM(0, 1) S3_TERM(1, 0) S3_TERM(0, 0)
  };
#undef S3_TERM

  // ---- epilogue of the tile whose last chunk was just multiplied: coordinates decoded in that chunk's load phase, stores one phase later
  int l_comp = l_first, c_kc = 0;
  int e_z = 0, e_m0 = 0, e_n0 = 0;
  bool epi_pending = false;
  // (the per-channel vectors are fetched here, channel fragment by channel fragment -- L2-resident, and the whole epilogue runs beside the partner
  // group's MFMAs; holding them over the tile's last chunk as the one-tile kernel does costs 32 registers and spills)
  auto epilogue = [&]() __attribute__((always_inline)) {
    const long y_base = (long)e_z * M * p.y_ld;
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = e_n0 + wn * WTN + fn * 16 + fg * 4;
      float4 bias_r = make_float4(0.f, 0.f, 0.f, 0.f), scale_r = make_float4(1.f, 1.f, 1.f, 1.f);
      if constexpr (!BARE) {
        __builtin_amdgcn_sched_barrier(0);                 // one channel fragment at a time: hoisting every residual load of the tile costs
                                                           // registers the live fragment set does not leave
        if (n < p.Cout) {
          if (p.bias) bias_r = *reinterpret_cast<const float4*>(p.bias + n);
          if (p.scale) scale_r = *reinterpret_cast<const float4*>(p.scale + n);
        }
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int m = e_m0 + wm * WTM + fm * 16 + fr;
        if (m < M && n < p.Cout && !S3P_DBG(8)) {
          if constexpr (BARE) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + y_base + (long)m * p.y_ld + n) = acc[fn][fm];
          } else {
            float v[4] = {acc[fn][fm][0] + bias_r.x, acc[fn][fm][1] + bias_r.y, acc[fn][fm][2] + bias_r.z, acc[fn][fm][3] + bias_r.w};
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
            v[0] *= scale_r.x; v[1] *= scale_r.y; v[2] *= scale_r.z; v[3] *= scale_r.w;
            if (p.res) {
              const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (long)m * p.res_ld + n);
              v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            }
            if (p.res2) {
              const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res2) + (long)m * p.res2_ld + n);
              v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            }
            if (p.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + y_base + (long)m * p.y_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
            else store_split3(reinterpret_cast<bf16_t*>(p.y) + split3_at(m, n, p.y_ld, (p.korder & 8) ? M : 0), p.y_bstride, v);
          }
        }
        acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    epi_pending = false;
  };

  // ---- the chunk stream (schedule and hazards: see the ping-pong branch of gemm_split3_kernel; tests/test_split3_schedule_model_cpu.py
  // replays this control flow, tile switches included)
#pragma nounroll
  for (int i = 0; i < NS; ++i)
    if (i < chunks) issue();
  if (chunks > 2) vm_wait<PPW>();                       // chunks 0 and 1 landed
  else vm_wait<0>();
  lds_barrier();
  Frags fa, fb;
  read_frags(fa);
  lds_barrier();
  const bool grp_b = wave >= NW / 2;
  if (grp_b) plain_barrier();                           // one phase behind
  // phase M(g): [stores of the tile that ended with chunk g-1], DMA of chunk g+3, fragments of chunk g+1, wait for this wave's pieces of g+2;
  // phase C(g): the 48 MFMAs of chunk g out of registers
#ifdef PF_S3_DBG
  // timeline build (tools/persist_probe.py timeline): waves 0 and 4 of block 0 stamp s_memtime at the seams of chunks 64 .. 95 into p.res2
  // (8 stamps per chunk and wave: phase start, after the epilogue branch, after the DMA issue, after the fragment reads were issued, after the
  // vmcnt wait, after the LDS barrier, after the MFMAs, after the closing barrier)
  unsigned long long* tl = BARE ? nullptr : nullptr;
  if constexpr (BARE) tl = reinterpret_cast<unsigned long long*>(const_cast<void*>(p.res2));
  const bool stamp_w = tl && blockIdx.x == 0 && (wave == 0 || wave == 4);
#define S3_STAMP(g, k)                                                                                      \
  if (stamp_w && (g) >= 64 && (g) < 96) {                                                                   \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                             \
    if (lane == 0) tl[(((wave >> 2) * 32 + ((g) - 64)) << 3) + (k)] = t_;                                    \
  }
#else
#define S3_STAMP(g, k)
#endif
  auto pp_chunk = [&](const Frags& cf, Frags& nf, int g) __attribute__((always_inline)) {
    S3_STAMP(g, 0)
    if (epi_pending) epilogue();                        // beside the partner group's MFMAs
    if (c_kc == nk - 1) {                               // chunk g ends a tile
      decode(l_comp, e_z, e_m0, e_n0);
      l_comp += nb;
    }
    S3_STAMP(g, 1)
    if (g + 3 < chunks) issue();
    S3_STAMP(g, 2)
    if (g + 1 < chunks) read_frags(nf);
    S3_STAMP(g, 3)
    if (g + 3 < chunks) vm_wait<PPW>();
    else if (g + 2 < chunks) vm_wait<0>();
    S3_STAMP(g, 4)
    lds_barrier();
    S3_STAMP(g, 5)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    multiply(cf);
    __builtin_amdgcn_s_setprio(0);
    S3_STAMP(g, 6)
    if (++c_kc == nk) { c_kc = 0; epi_pending = true; }
    plain_barrier();
    S3_STAMP(g, 7)
  };
#pragma nounroll
  for (int g = 0; g < chunks; g += 2) {
    pp_chunk(fa, fb, g);
    if (g + 1 < chunks) pp_chunk(fb, fa, g + 1);
  }
  if (!grp_b) plain_barrier();
  epilogue();
}

// ---- PERSISTENT 192 x 192 form (round 4, late) ---------------------------------------------------------------------------------------------------
// Why: the 128 x 128 kernel above is bounded by its LOAD phase, not by the matrix pipe.  Timeline of the dominant launch (tools/persist_probe.py
// timeline, profiles/r4_persist_timeline.md): a wave's load phase = 6 LDS-DMA pieces (~95 cycles each wherever they are issued -- moving them
// among the MFMAs was measured zero-sum, see above) + 18 ds_read_b128 (~16 each) ~= 900 cycles against 48 MFMAs = 768 in the partner's
// compute phase, so a phase lasts ~1000-1050 cycles and the matrix pipe is busy 73 % of the time (PMC: SQ_VALU_MFMA_BUSY_CYCLES 0.735).
// A 192 x 192 tile (eight waves of 48 tokens x 96 channels) does 108 MFMAs per wave and chunk (1728 cycles) for 9 pieces + 27 fragment reads
// (~1300): 2.25x the products for 1.5x the operand bytes, the compute phase is the long one, a third less L2 / HBM operand traffic per product,
// and N = 544 pads to 576 (5.9 % dead columns) instead of 640 (15 %); 768 = 4 x 192 stays exact.
// LDS: one stage = 3 planes x (192 + 192) rows x 64 B = 72 KiB, so the ring has TWO slots (144 KiB) and the fragments are SINGLE-buffered in
// registers (27 x 4 + 72 accumulators; the ping-pong partner covers the LDS latency).  Schedule in global phases (barrier to barrier); group A
// (waves 0-3, loads the X planes) runs M_A(g) in phase 2g and C_A(g) in phase 2g+1, group B (waves 4-7, the W planes) M_B(g) in 2g+1 and C_B(g) in
// 2g+2; chunk g lives in slot g & 1:
//   phase 2h   : EVERY wave issues its pieces of chunk h+1 into slot (h+1) & 1 -- group A at the head of its load phase M_A(h), group B between the
//                MFMAs of C_B(h-1).  The slot's previous occupant, chunk h-1, was read by A in phase 2h-2 and by B in phase 2h-1, each with
//                lgkmcnt(0) before the phase's closing barrier.
//   phase 2h+1 : every wave waits for its own pieces of chunk h+1 (vmcnt(0)) before the closing barrier -- A at the end of C_A(h), B at the end of
//                M_B(h); A reads chunk h+1 in phase 2h+2, B in 2h+3.
// (tests/test_split3_schedule_model_cpu.py persist192 replays this control flow and asserts both rules for every chunk count.)
// BL (round 5): group B issues its W pieces of chunk g+1 at the HEAD OF ITS OWN LOAD PHASE M_B(g) (phase 2g+1) instead of between the MFMAs of its
// compute phase one phase earlier.  The register-resident microbenchmark (tools/mfma_ceiling.hip, profiles/r5_mfma_ceiling.md) shows the barrier
// skeleton alone keeps the pipe 99.5 % busy; what idles it in the product kernel is every non-MFMA instruction of the ONE wave per SIMD that is
// allowed to multiply -- a DMA piece stalls that wave ~100 cycles.  The slot (g+1) & 1 is free from phase 2g on (chunk g-1: read by A in 2g-2, by B
// in 2g-1), group A reads chunk g+1 in phase 2g+2, so B's pieces have the rest of phase 2g+1 to land: they are the W planes (filters / weights,
// L2-resident: 250-400 cycles), B waits vmcnt(0) at the end of that phase exactly as before.  Compute phases of both groups are then MFMA-only.
template <bool BARE, bool BL>
__global__ __launch_bounds__(512) void gemm_split3_persist192_kernel(const pf_conv_params p, int mt, int nt, int gm, int total, int flags) {
  constexpr int BM = 192, BN = 192, WM = 4, WN = 2, NP = 3, NS = 2;
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int ROWS = NP * (BM + BN), PIECES = ROWS / 16, PPW = PIECES / NW;
  constexpr int STAGE = ROWS * 64;
  constexpr int NMF = 6 * FN * FM, MPP = NMF / PPW;            // MFMAs per chunk and wave; MFMAs between two pieces of group B
  static_assert(PPW * 16 * WM == NP * BM && WM * 2 == NW && BM == BN && NMF % PPW == 0, "waves 0..3 stage the X planes, waves 4..7 the W planes");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = p.B * p.OH * p.OW;
  const int nk = p.Cin / 32;

  // ---- this block's tiles (same walk as the 128 x 128 kernel)
  const int G = (int)gridDim.x, xcd = (int)blockIdx.x & 7, nb = (G - xcd + 7) >> 3;
  const int lo = (int)((long)total * xcd / 8), hi = (int)((long)total * (xcd + 1) / 8);
  const int l_first = lo + ((int)blockIdx.x >> 3);
  if (l_first >= hi) return;
  const int my_tiles = (hi - l_first + nb - 1) / nb;
  const int chunks = my_tiles * nk;
  const int per_plane = mt * nt, per_group = gm * nt;
  auto decode = [&](int l, int& z, int& m0, int& n0) __attribute__((always_inline)) {
    z = l / per_plane;
    const int r = l - z * per_plane;
    if (gm == 0) {                                        // channel tile fastest (tile_group: nt does not divide an XCD's 32 blocks)
      const int t = r / nt;
      m0 = t * BM;
      n0 = (r - t * nt) * BN;
      return;
    }
    const int group = r / per_group, in_g = r - group * per_group, first = group * gm;
    const int gsz = min(mt - first, gm);
    const int tn = in_g / gsz;
    m0 = (first + in_g - tn * gsz) * BM;
    n0 = tn * BN;
  };

  // ---- loader: wave w moves pieces w*PPW .. +PPW-1 of a stage [X h | X m | X l | W h | W m | W l] (12 pieces per plane)
  const bool is_x = wave < WM;
  const bool kmaj = (p.korder & (is_x ? 2 : 4)) != 0;
  const int lim = is_x ? M : p.w_rows;
  const int ld_b = kmaj ? 64 : (is_x ? p.x_ld : p.Kpad) * 2;
  const unsigned adv_b = kmaj ? (unsigned)lim * 64u : 64u;
  const unsigned long long op_base = is_x ? (unsigned long long)p.x : (unsigned long long)p.w;
  const unsigned long long pl_b = (unsigned long long)(is_x ? p.x_bstride : p.w_bstride) * 2;
  const unsigned long long z_b = (unsigned long long)lim * ((is_x ? p.x_ld : p.Kpad) * 2);
  unsigned long long sbase[PPW];
  unsigned voff[PPW];
  auto setup_loader = [&](int l) __attribute__((always_inline)) {
    int z, m0, n0;
    decode(l, z, m0, n0);
#ifdef PF_S3_DBG           // traffic experiment (results wrong by construction): p.pad bit 4 = every tile reads the X rows of token tile 0 of plane 0
    if (p.pad & 16) { m0 = 0; if (is_x) z = 0; }
#endif
    const int origin = is_x ? m0 : n0;
    const unsigned long long tb = op_base + (unsigned long long)z * z_b + (unsigned long long)origin * ld_b;
    const int last = lim - 1 - origin;

    int lane_o = lane;                                    // (opaque copy: the per-piece row constants are recomputed at every tile switch instead of
    asm volatile("" : "+v"(lane_o));                      //  being hoisted out of the chunk loop, where nine more live registers would spill)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = (wave & (WM - 1)) * PPW + i;                                // piece within this operand's three planes
      const int row = pc * 16 + (lane_o >> 2);
      const int j = (lane_o & 3) ^ ((row >> 1) & 3);                             // (BM is a multiple of 8: same swizzle as the stage row)
      const int pl = pc / (BM / 16), r = row - pl * BM;
      sbase[i] = tb + pl * pl_b;
      voff[i] = (unsigned)(min(r, last) * ld_b + j * 16);
    }
  };
  const unsigned smem_base = lds_addr(smem);
  int l_load = l_first, l_kc = -1, s_issue = NS - 1;
  unsigned dst_cur = 0;
  setup_loader(l_load);
  auto begin_issue = [&]() __attribute__((always_inline)) {     // cursor of the next chunk to load: ring slot, chunk within the tile, tile switch
    s_issue ^= 1;
    if (++l_kc == nk) {
      l_kc = 0;
      l_load += nb;
      if (l_load < hi) setup_loader(l_load);
    }
    dst_cur = __builtin_amdgcn_readfirstlane(smem_base + s_issue * STAGE + wave * (PPW * 1024));
  };
  auto piece = [&](int i) __attribute__((always_inline)) {
    glds16s(voff[i], sbase[i], dst_cur + i * 1024);
    sbase[i] += adv_b;
  };
  // (Measured and NOT kept, round 5: an L2 prefetch of the X planes one chunk ahead of the pieces -- two dword LDS-DMA loads per wave and chunk touching
  // every 128-byte line of the next chunk's rows.  0.98-1.01x on the Winograd GEMMs (profiles/r5_flags_sweep.md, flags bit 0); the ~300-cycle vmcnt(0)
  // wait the s_memtime timeline shows after the MFMAs is the completion of the STAMP's own store, not DMA latency.)
  auto issue = [&]() __attribute__((always_inline)) {
    begin_issue();
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece(i);
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int slot = (fg ^ ((fr >> 1) & 3)) << 4;
  const int x_off = (wm * WTM + fr) * 64 + slot;
  // NARROW LAST COLUMN TILE (round 5): when the tile has nv < BN valid columns (N = 544: 160 of 192) the two wave columns share them evenly --
  // cur_nfn = ceil(ceil(nv / 16) / WN) fragments of 16 columns per wave (5 instead of 6: 90 MFMAs per chunk and wave instead of 108, on BOTH wave columns,
  // where a fixed 96 + 96 split leaves wave column 0 at 108 and wave column 1 multiplying 32 dead columns) -- the fragments above cur_nfn are neither read
  // nor multiplied nor stored.  Same products in the same order for every output element (bit-identical).  Never fewer than NFN_MIN = 3 (the guard is a
  // wave-uniform branch only on the fragments above it).
  constexpr int NFN_MIN = 3;
  int cur_nfn = FN, cur_wtn = WTN, e_nfn = FN, e_wtn = WTN;
  int w_off = NP * BM * 64 + (wn * WTN + fr) * 64 + slot;
  auto tile_shape = [&](int n0) __attribute__((always_inline)) {
    if (!(flags & 8)) return;                            // (A/B: PF_S3_FLAGS without bit 3 keeps the fixed 96 + 96 split)
    const int nv = min(BN, p.Cout - n0);
    cur_nfn = min(FN, max(NFN_MIN, ((nv + 15) / 16 + WN - 1) / WN));
    cur_wtn = cur_nfn * 16;
    w_off = NP * BM * 64 + (wn * cur_wtn + fr) * 64 + slot;
  };
  struct Frags { uint4 w[NP][FN], x[NP][FM]; };
  Frags f;
  int s_read = 0;
  auto read_frags = [&]() __attribute__((always_inline)) {
    const char* S = smem + s_read * STAGE;
    s_read ^= 1;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
        if (fn < NFN_MIN || fn < cur_nfn) f.w[pl][fn] = *reinterpret_cast<const uint4*>(S + w_off + pl * (BN * 64) + fn * 1024);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) f.x[pl][fm] = *reinterpret_cast<const uint4*>(S + x_off + pl * (BM * 64) + fm * 1024);
    }
  };
  // six partial products, smallest first (the order of gemm_split3_kernel: same bits); ISS: one DMA piece ahead of every MPP-th MFMA
#define S3_TERM192(TI, PW, PX)                                                                                                  \
  _Pragma("unroll") for (int fn = 0; fn < FN; ++fn) _Pragma("unroll") for (int fm = 0; fm < FM; ++fm) {                      \
    if (ISS && ((TI) * FN * FM + fn * FM + fm) % MPP == 0) {                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                       \
      piece(((TI) * FN * FM + fn * FM + fm) / MPP);                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                                       \
    }                                                                                                                          \
    if (fn < NFN_MIN || fn < cur_nfn)                                                                                          \
      acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.w[PW][fn]), __builtin_bit_cast(bf16x8, f.x[PX][fm]), \
                                                            acc[fn][fm], 0, 0, 0);                                             \
  }
  auto multiply = [&](auto with_issue) __attribute__((always_inline)) {
    constexpr bool ISS = decltype(with_issue)::value;
    S3_TERM192(0, 0, 2) S3_TERM192(1, 2, 0) S3_TERM192(2, 1, 1) S3_TERM192(3, 0, 1) S3_TERM192(4, 1, 0) S3_TERM192(5, 0, 0)
  };
#undef S3_TERM192

  int l_comp = l_first, c_kc = 0;
  int e_z = 0, e_m0 = 0, e_n0 = 0;
  bool epi_pending = false;
  // Epilogue on EXCHANGED fragments (round 5).  A fragment's store instruction would write 16 rows x 64 bytes -- sixteen half cache lines; the two channel
  // fragments fn, fn+1 of a row are the two halves of ONE 128-byte line.  Lanes fr < 8 trade their fn+1 fragment for the fn fragment of lane fr + 8 (DPP
  // row_ror:8, four moves); each lane then owns rows fr & 7 and (fr & 7) + 8 of ONE 4-channel group.  Everything after the accumulation is elementwise,
  // so bias -> act -> scale -> residual(s) run on the exchanged values (a single bias / scale vector per lane; residual loads are whole lines too):
  //   float32 output: two 16-byte stores per fragment pair, eight FULL lines each;
  //   plane output, chunk-major (the ViT block's qkv / fc1): a lane's share of a plane row is 8 bytes.  v_permlane16_swap_b32 trades words between lanes
  //   16 apart = the adjacent 4-channel group of the same row: swap(h, m) leaves the even group with 16 bytes of plane h and the odd group with 16 bytes
  //   of plane m; the l words of the lane's two rows pair up the same way -- three full 16-byte store instructions where six 8-byte ones were;
  //   plane output, row-major (tests): plain 8-byte stores.
  // Same bits to the same addresses as a fragment-by-fragment epilogue.
  auto epilogue = [&]() __attribute__((always_inline)) {
    const long y_base = (long)e_z * M * p.y_ld;
    int fr = lane & 15, fg = lane >> 4;                  // (re-derived behind an opaque barrier: everything computed from them stays INSIDE the
    asm volatile("" : "+v"(fr), "+v"(fg));               //  epilogue -- hoisted out of the chunk loop these values spill the loader's piece offsets)
    const bool lo = fr < 8, oddg = fg & 1;
#pragma unroll
    for (int fn = 0; fn < FN; fn += 2) {
      __builtin_amdgcn_sched_barrier(0);                 // one fragment pair at a time (register pressure)
      const int fsel = fn + (lo ? 0 : 1);                // the fragment whose values this lane stores
      const int n = e_n0 + wn * e_wtn + fsel * 16 + fg * 4;
      const bool nok = n < p.Cout && fsel < e_nfn;
      float4 bias_r = make_float4(0.f, 0.f, 0.f, 0.f), scale_r = make_float4(1.f, 1.f, 1.f, 1.f);
      if constexpr (!BARE) {
        if (nok) {
          if (p.bias) bias_r = *reinterpret_cast<const float4*>(p.bias + n);
          if (p.scale) scale_r = *reinterpret_cast<const float4*>(p.scale + n);
        }
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const f32x4 a = acc[fn][fm], b = acc[fn + 1][fm];
        acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[fn + 1][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
        float v[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float snd = lo ? b[r] : a[r];
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, snd), 0x128, 0xf, 0xf, false));   // row_ror:8
          v[0][r] = lo ? a[r] : recv;
          v[1][r] = lo ? recv : b[r];
        }
        const int m1 = e_m0 + wm * WTM + fm * 16 + (fr & 7);
        if constexpr (BARE) {
          float* yb = reinterpret_cast<float*>(p.y) + y_base;
          if (nok && m1 < M) *reinterpret_cast<float4*>(yb + (long)m1 * p.y_ld + n) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
          if (nok && m1 + 8 < M) *reinterpret_cast<float4*>(yb + (long)(m1 + 8) * p.y_ld + n) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
        } else {
          const float bs[4] = {bias_r.x, bias_r.y, bias_r.z, bias_r.w}, sc[4] = {scale_r.x, scale_r.y, scale_r.z, scale_r.w};
          const int ne = n - (oddg ? 4 : 0);              // chunk-major plane output: the pair's 8 channels start at the even group
          uint32_t lw[2][2];                              // l words of the two rows (plane output)
#pragma unroll
          for (int h = 0; h < 2; ++h) {                   // one row at a time (register pressure)
            __builtin_amdgcn_sched_barrier(0);
            const int m = m1 + 8 * h;
            const bool ok = nok && m < M;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[h][r] += bs[r];
            if (p.act == PF_ACT_RELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[h][r] = fmaxf(v[h][r], 0.f);
            } else if (p.act == PF_ACT_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[h][r] = gelu_erf(v[h][r]);
            } else if (p.act == PF_ACT_SOFTPLUS) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[h][r] = softplus20(v[h][r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[h][r] *= sc[r];
            if (p.res && ok) {
              const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (long)m * p.res_ld + n);
              v[h][0] += q.x; v[h][1] += q.y; v[h][2] += q.z; v[h][3] += q.w;
            }
            if (p.res2 && ok) {
              const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res2) + (long)m * p.res2_ld + n);
              v[h][0] += q.x; v[h][1] += q.y; v[h][2] += q.z; v[h][3] += q.w;
            }
            if (p.out_f32) {
              if (ok) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + y_base + (long)m * p.y_ld + n) = make_float4(v[h][0], v[h][1], v[h][2], v[h][3]);
            } else if (!(p.korder & 8)) {
              if (ok) store_split3(reinterpret_cast<bf16_t*>(p.y) + split3_at(m, n, p.y_ld, 0), p.y_bstride, v[h]);
            } else {
              uint32_t h0, m0, h1, mm1;
              split3_pair(v[h][0], v[h][1], h0, m0, lw[h][0]);
              split3_pair(v[h][2], v[h][3], h1, mm1, lw[h][1]);
              // (every lane executes the swaps; only the stores are guarded)
              const auto s0 = __builtin_amdgcn_permlane16_swap(h0, m0, false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(h1, mm1, false, false);
              if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y) + split3_at(m, ne, p.y_ld, M) + (oddg ? p.y_bstride : 0)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
          }
          if (!p.out_f32 && (p.korder & 8)) {
            const auto t0 = __builtin_amdgcn_permlane16_swap(lw[0][0], lw[1][0], false, false);
            const auto t1 = __builtin_amdgcn_permlane16_swap(lw[0][1], lw[1][1], false, false);
            const int ml = m1 + (oddg ? 8 : 0);
            if (nok && ml < M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y) + split3_at(ml, ne, p.y_ld, M) + 2 * p.y_bstride) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
          }
        }
      }
    }
    epi_pending = false;
  };
  auto tile_cursor = [&]() __attribute__((always_inline)) {     // head of a load phase: stores of the finished tile, coordinates of the ending one
    if (epi_pending) epilogue();                        // beside the partner group's MFMAs
    if (c_kc == 0) {                                    // a tile begins: its column shape (narrow last column tile)
      int z, m0, n0;
      decode(l_comp, z, m0, n0);
      tile_shape(n0);
    }
    if (c_kc == nk - 1) {
      decode(l_comp, e_z, e_m0, e_n0);
      e_nfn = cur_nfn;
      e_wtn = cur_wtn;
      l_comp += nb;
    }
  };
  auto tile_advance = [&]() __attribute__((always_inline)) {
    if (++c_kc == nk) { c_kc = 0; epi_pending = true; }
  };

#ifdef PF_S3_DBG
  // timeline build (tools/persist_probe.py timeline192): waves 0 and 4 of block 0 stamp s_memtime at the seams of chunks 64 .. 95 into p.res2
  // (8 stamps per chunk and wave, in program order; see the loops below)
  unsigned long long* tl = nullptr;
  if constexpr (BARE) tl = reinterpret_cast<unsigned long long*>(const_cast<void*>(p.res2));
  const bool stamp_w = tl && blockIdx.x == 0 && (wave == 0 || wave == 4);
#define S3_STAMP192(g, k)                                                                                     \
  if (stamp_w && (g) >= 64 && (g) < 96) {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
    if (lane == 0) tl[(((wave >> 2) * 32 + ((g) - 64)) << 3) + (k)] = t_;                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
#else
#define S3_STAMP192(g, k)
#endif
  // (Measured and NOT kept, round 5: staggering the block starts by up to one tile time so that the CUs do not reach their tile ends -- 147 KB of stores
  // each -- at the same moments.  No gain (profiles/r5_flags_sweep_stagger.md): the store burst of a tile, ~4700 cycles per wave group in the s_memtime
  // timeline against a ~1870-cycle partner phase, is bound by the CU's own store issue (~16 B/clk/CU), not by the chip-wide write bandwidth.)
  // ---- the chunk stream
  issue();                                              // chunk 0
  vm_wait<0>();
  lds_barrier();
  if (wave < NW / 2) {
    // group A: M_A(g) in phase 2g, C_A(g) in phase 2g+1
#pragma nounroll
    for (int g = 0; g < chunks; ++g) {
      S3_STAMP192(g, 0)
      tile_cursor();
      S3_STAMP192(g, 1)
      if (g + 1 < chunks) issue();                      // chunk g+1, phase 2g
      S3_STAMP192(g, 2)
      read_frags();
      S3_STAMP192(g, 3)
      lds_barrier();
      S3_STAMP192(g, 4)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      multiply(std::false_type{});
      __builtin_amdgcn_s_setprio(0);
      S3_STAMP192(g, 5)
      tile_advance();
      vm_wait<0>();                                     // this wave's pieces of chunk g+1 have landed (phase 2g+1)
      S3_STAMP192(g, 6)
      plain_barrier();
      S3_STAMP192(g, 7)
    }
    plain_barrier();                                    // group B's last compute phase
  } else if constexpr (BL) {
    // group B, pieces from its own load phase: M_B(g) in phase 2g+1 issues chunk g+1, reads chunk g; C_B(g) in phase 2g+2 is MFMA-only
    plain_barrier();                                    // phase 0 (group A's first load phase)
#pragma nounroll
    for (int g = 0; g < chunks; ++g) {
      S3_STAMP192(g, 0)
      tile_cursor();
      S3_STAMP192(g, 1)
      if (g + 1 < chunks) issue();                      // W pieces of chunk g+1 -> slot (g+1) & 1, phase 2g+1
      S3_STAMP192(g, 2)
      read_frags();                                     // chunk g
      S3_STAMP192(g, 3)
      vm_wait<0>();                                     // they have landed before group A reads them in phase 2g+2
      S3_STAMP192(g, 4)
      lds_barrier();
      S3_STAMP192(g, 5)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      multiply(std::false_type{});
      __builtin_amdgcn_s_setprio(0);
      S3_STAMP192(g, 6)
      tile_advance();
      plain_barrier();
      S3_STAMP192(g, 7)
    }
  } else {
    // group B: one phase behind; M_B(g) in phase 2g+1, C_B(g) in phase 2g+2
    if (1 < chunks) issue();                            // chunk 1, phase 0
    plain_barrier();
    int g = 0;
#pragma nounroll
    for (; g + 2 < chunks; ++g) {
      tile_cursor();
      begin_issue();                                    // cursor of chunk g+2 (no memory operation: the pieces follow in the compute phase)
      read_frags();
      vm_wait<0>();                                     // pieces of chunk g+1 (issued in phase 2g)
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      multiply(std::true_type{});                       // chunk g, with the pieces of chunk g+2 (phase 2g+2)
      __builtin_amdgcn_s_setprio(0);
      tile_advance();
      plain_barrier();
    }
#pragma nounroll
    for (; g < chunks; ++g) {                           // the last two chunks: nothing left to issue
      tile_cursor();
      read_frags();
      vm_wait<0>();
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      multiply(std::false_type{});
      __builtin_amdgcn_s_setprio(0);
      tile_advance();
      plain_barrier();
    }
  }
  epilogue();
}

thread_local char g_err[200] = {0};
constexpr int PF_ERR_FALLBACK = -100;     // internal: the persistent launch is not legal for this grid, use the one-tile kernel

int cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  hipGetDevice(&dev);
  int c = cus[dev & 63].load(std::memory_order_relaxed);
  if (c <= 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cus[dev & 63].store(c, std::memory_order_relaxed);
  }
  return c;
}

// Tile order of the persistent kernels inside a transform point: the ~32 tiles an XCD works on at once are consecutive in the order, and they
// should be a patch whose X and W panels are shared through that XCD's L2.  nt (channel tiles) dividing 32: gm = 32 / nt token tiles x all channel
// tiles, column by column; nt > 6: 8 token tiles x 4 channel tiles.  nt = 3, 5, 6 (30-tile patches): a third of the blocks that share an X panel
// would fall into DIFFERENT iterations of the 32-block XCD (46 us apart: an L2 miss each) -- those run channel-tile-fastest instead (returned as
// gm = 0): the nt sharers of a token tile are adjacent in the order, all W panels (nt x 209 KB at K = 544) stay hot.  PF_S3_ORDER=0: the patches.
// Round 5: the same holds for the "aligned" nt (2, 4, 8, 16, 22 ...) -- the XCD ranges and per-plane tile counts are not multiples of 32, so the
// patches split 70 % of their panels too (host replay, tests/test_split3_schedule_model_cpu.py) -- measured per launch (profiles/r5_order_sweep_unbiased.md):
// fc1 1.07x, the 8296 x 1024 projection 1.04x, the N = 256 layers 1.03x, 768->768 (nt = 4) 1.01-1.02x; the image pass -1.0 / -3.0 / -2.2 ms in three
// interleaved A/Bs.  Channel tile fastest is therefore the default for every nt >= 2 (PF_S3_ORDER=1: the round-4 rule).
int tile_group(int nt) {
  const char* s = getenv("PF_S3_ORDER");
  const char o = s ? s[0] : '2';
  if (o == '2' && nt >= 2) return 0;                    // round 5: channel tile fastest for every nt (profiles/r5_order_sweep.md)
  if (o == '1' && (nt == 3 || nt == 5 || nt == 6)) return 0;       // the round-4 rule
  return nt <= 6 ? 32 / nt : 8;                         // PF_S3_ORDER=0: the patches
}

// persistent launch: 128 x 128 tiles, one block per CU (grid = a multiple of 8 so that every XCD has blocks)
int launch_persist(const pf_conv_params& p, hipStream_t st, int grid_cap) {
  constexpr int smem = 3 * 3 * (128 + 128) * 64;
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_persist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_persist_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    done.fetch_or(bit, std::memory_order_release);
  }
  const long M = (long)p.B * p.OH * p.OW;
  const int mt = (int)((M + 127) / 128), nt = (p.Cout + 127) / 128;
  const long total = (long)mt * nt * (p.batch > 1 ? p.batch : 1);
  const int gm = tile_group(nt);
  int grid = cu_count();
  if (grid_cap > 0 && grid_cap < grid) grid = grid_cap;                  // (pf_gemm_split3_ex: leave CUs to a concurrent HBM-bound stream)
  if (const char* s = getenv("PF_S3_GRID")) grid = atoi(s);              // (tests: fewer blocks than CUs = more tiles per block; read per call)
  if (grid > total) grid = (int)total;
  grid &= ~7;                                                            // (8 XCDs on gfx950: the walk gives every XCD a contiguous tile range)
  if (total > 0x7fffffffL) return PF_ERR_ARG;
  if (grid < 8) return PF_ERR_FALLBACK;                                  // (a PF_S3_GRID below 8 / not a number: the caller runs the one-tile kernel)
#ifdef PF_S3_DBG
  const bool bare = !p.bias && !p.scale && !p.res && p.act == PF_ACT_NONE && p.out_f32;          // (res2 = the timeline buffer)
#else
  const bool bare = !p.bias && !p.scale && !p.res && !p.res2 && p.act == PF_ACT_NONE && p.out_f32;
#endif
  if (bare) hipLaunchKernelGGL(gemm_split3_persist_kernel<true>, dim3((unsigned)grid), dim3(512), smem, st, p, mt, nt, gm, (int)total);
  else hipLaunchKernelGGL(gemm_split3_persist_kernel<false>, dim3((unsigned)grid), dim3(512), smem, st, p, mt, nt, gm, (int)total);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

// persistent 192 x 192 launch (one block per CU, 144 KiB of LDS)
int launch_persist192(const pf_conv_params& p, hipStream_t st, int grid_cap) {
  constexpr int smem = 2 * 3 * (192 + 192) * 64;
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_persist192_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_persist192_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_persist192_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split3_persist192_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    done.fetch_or(bit, std::memory_order_release);
  }
  const long M = (long)p.B * p.OH * p.OW;
  const int mt = (int)((M + 191) / 192), nt = (p.Cout + 191) / 192;
  const long total = (long)mt * nt * (p.batch > 1 ? p.batch : 1);
  const int gm = tile_group(nt);
  int grid = cu_count();
  if (grid_cap > 0 && grid_cap < grid) grid = grid_cap;                  // (pf_gemm_split3_ex: leave CUs to a concurrent HBM-bound stream)
  if (const char* s = getenv("PF_S3_GRID")) grid = atoi(s);              // (tests: fewer blocks than CUs = more tiles per block; read per call)
  if (grid > total) grid = (int)total;
  grid &= ~7;                                                            // (8 XCDs on gfx950: the walk gives every XCD a contiguous tile range)
  if (total > 0x7fffffffL) return PF_ERR_ARG;
  if (grid < 8) return PF_ERR_FALLBACK;                                  // (a PF_S3_GRID below 8 / not a number: the caller runs the one-tile kernel)
#ifdef PF_S3_DBG
  const bool bare = !p.bias && !p.scale && !p.res && p.act == PF_ACT_NONE && p.out_f32;          // (res2 = the timeline buffer)
#else
  const bool bare = !p.bias && !p.scale && !p.res && !p.res2 && p.act == PF_ACT_NONE && p.out_f32;
#endif
  const char* bls = getenv("PF_S3_BLOAD");                               // (A/B: 0 = group B issues between its MFMAs, the round-4 schedule)
  const bool bl = !(bls && bls[0] == '0');
  int flags = 8;                                                         // bit 3: narrow last column tile shared evenly by the wave columns (A/B: PF_S3_FLAGS=0)
  if (const char* s = getenv("PF_S3_FLAGS")) flags = atoi(s);
  if (bare && bl) hipLaunchKernelGGL((gemm_split3_persist192_kernel<true, true>), dim3((unsigned)grid), dim3(512), smem, st, p, mt, nt, gm, (int)total, flags);
  else if (bare) hipLaunchKernelGGL((gemm_split3_persist192_kernel<true, false>), dim3((unsigned)grid), dim3(512), smem, st, p, mt, nt, gm, (int)total, flags);
  else if (bl) hipLaunchKernelGGL((gemm_split3_persist192_kernel<false, true>), dim3((unsigned)grid), dim3(512), smem, st, p, mt, nt, gm, (int)total, flags);
  else hipLaunchKernelGGL((gemm_split3_persist192_kernel<false, false>), dim3((unsigned)grid), dim3(512), smem, st, p, mt, nt, gm, (int)total, flags);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

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

// Which kernel runs a (validated) call on a chip of `cus` compute units -- the whole dispatch rule in one place (pf_gemm_split3_route exposes it to
// the CPU tests).  The PF_S3_* switches are A/B and test knobs, read per call.
//   TILE64      64 x 128 tiles, two-slot ring (two blocks per CU): the token x channel grid does not fill the chip once with 128 x 128 tiles
//   TILE128     one 128 x 128 tile per block, ping-pong (profiles/r3_split3_pingpong.log: wins from one full round of tiles on)
//   PERSIST128  from two rounds of tiles on: one resident block per CU walks its tiles, chunk stream continuous across tiles (round 4)
//   PERSIST192  the same walk over 192 x 192 tiles on a two-slot ring when that costs less: rounds of tiles per CU x cost of a tile, a 192-tile
//               being 2.25x the products of a 128-tile run ~8 % faster (profiles/r4_t192_sweep.md: 768->768 1.06x, exact in both tilings) = 2.1,
//               plus half a round for the coarser tail (768->768 @ 8x56x74, 7 rounds of 192-tiles: 0.455 ms against 0.415 on 15 rounds of 128-tiles).
//               N = 544 pads to 576 instead of 640 columns (1.15x), 768 / 3072 / 4096 gain 1.05-1.12x; N <= 256 and the 8296 x 1024
//               projections (264 tiles on 256 CUs) stay on 128 x 128.  PF_S3_T192: 0 = never, 2 = wherever legal.
int split3_route(const pf_conv_params& p, int cus) {
  int force = 0;
  if (const char* s = getenv("PF_S3_TILE_NOW")) force = atoi(s);
  const long M = (long)p.B * p.OH * p.OW;
  const long t128 = ((M + 127) / 128) * ((p.Cout + 127) / 128);
  const long planes = p.batch > 1 ? p.batch : 1;
  // (tiles of ALL planes: the 36 transform points of a small Winograd layer -- 768->768 @ 8x56x74 is 102 tiles per point, 3672 in the launch --
  // used to count per plane and fell to the 64 x 128 kernel: 0.51 ms against 0.40 through the persistent walk)
  if (force ? force == 64 : t128 * planes < cus) return PF_S3_ROUTE_TILE64;
  const char* ps = getenv("PF_S3_PERSIST");
  if (!(ps && ps[0] == '0') && p.Cin >= 96 && (t128 * planes >= 2L * cus || (ps && ps[0] == '2')) && t128 * planes >= 8) {
    const char* ts = getenv("PF_S3_T192");
    const long t192 = ((M + 191) / 192) * ((p.Cout + 191) / 192);
    const long cost128 = ((t128 * planes + cus - 1) / cus) * 100, cost192 = ((t192 * planes + cus - 1) / cus) * 210 + 50;   // (+ half a round: coarser tail)
    const bool want = ts && ts[0] == '2' ? true : (ts && ts[0] == '0' ? false : cost192 < cost128);
    return want && t192 * planes >= 8 ? PF_S3_ROUTE_PERSIST192 : PF_S3_ROUTE_PERSIST128;
  }
  return PF_S3_ROUTE_TILE128;
}

}  // namespace

extern "C" int pf_gemm_split3_route(const pf_conv_params* p, int cus) {
  if (!p || cus <= 0) return -1;
  return split3_route(*p, cus);
}

namespace {
}  // namespace

extern "C" int pf_gemm_split3(const pf_conv_params* p, void* stream) { return pf_gemm_split3_ex(p, 0, stream); }

// grid_cap > 0: the persistent kernels launch at most that many blocks (one per CU; rounded down to a multiple of 8), the remaining CUs stay free
// for kernels of other streams -- the HBM-bound Winograd transforms of the other tile batch (csrc/winograd.hip run_split3).  0 = every CU.
extern "C" int pf_gemm_split3_ex(const pf_conv_params* p, int grid_cap, void* stream) {
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
  else if (((p->korder & 2) && p->x_ld != p->Cin) || ((p->korder & 4) && p->Kpad != p->Cin) || (p->korder & ~14)) e = "chunk-major operands are dense: x_ld == Kpad == Cin";
  else if ((p->korder & 8) && (p->out_f32 || p->y_ld != p->Cout || p->Cout % 32)) e = "chunk-major output: three planes, y_ld == Cout, Cout % 32 == 0";
  else if (((p->korder & 2) && (long)p->B * p->OH * p->OW * 64 >= (1L << 31)) || ((p->korder & 4) && (long)p->w_rows * 64 >= (1L << 31)))
    e = "too many rows for a chunk-major slab";
#ifdef PF_S3_DBG
  else if (p->batch > 1 && (!p->out_f32 || p->bias || p->scale || p->res || p->batch > 65535)) e = "batched planes: float32 output, no epilogue";
#else
  else if (p->batch > 1 && (!p->out_f32 || p->bias || p->scale || p->res || p->res2 || p->batch > 65535)) e = "batched planes: float32 output, no epilogue";
#endif
  if (e) return PF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (split3_route(*p, cu_count())) {
    case PF_S3_ROUTE_TILE64: return launch<64, 128, 2, 2, 2>(*p, st);
    case PF_S3_ROUTE_PERSIST192: {
      const int rc = launch_persist192(*p, st, grid_cap);
      if (rc != PF_ERR_FALLBACK) return rc;
      break;
    }
    case PF_S3_ROUTE_PERSIST128: {
      const int rc = launch_persist(*p, st, grid_cap);
      if (rc != PF_ERR_FALLBACK) return rc;
      break;
    }
    default: break;
  }
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
  pd.korder = 0;
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
