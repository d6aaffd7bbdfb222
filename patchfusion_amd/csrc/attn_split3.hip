// ViT attention in split precision, version 2 (round 6): the kernel that replaces vit_attention_split3_kernel (csrc/vit.hip) on the float32 headline
// path.  Same arithmetic, operand order and results (bit-identical: tests/op_checks.py vit_attention_split3_v2) -- softmax(q k^T / 8) v per head,
// head_dim 64 (dinov2/layers/attention.py:49-62), q / k / v and the output as three bf16 planes (x = h + m + l, csrc/gemm_split3.hip), both matrix
// products as the six leading partial products on v_mfma_f32_16x16x32_bf16 with float32 accumulation, float32 online softmax on base-2 logits.
//
// What version 1 was bound by (profiles/r4_attention_swizzle.md): every wave owned 16 queries, so each 16-byte K / V^T fragment read fed
// ONE MFMA per plane pair; K and V were staged through registers (48 per lane) with transposing 4-byte LDS stores (14 % of the LDS cycles still
// bank conflicts), two barriers around a serial load -> store -> compute sequence, MFMA pipe busy 41 %.  Version 2:
//   * a wave owns 16 QG queries (QG = 2 on the 8-tile launches): every K / V fragment read feeds QG MFMAs -- half the LDS reads per MFMA;
//   * K and V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no staging registers, no LDS stores, no
//     VALU); the DMA writes lane-linearly, so both bank swizzles are applied on the SOURCE side (which 16-byte chunk of the 128-byte row a lane
//     fetches);
//   * V stays ROW-major [key][d] in LDS; the V^T fragments of  O^T = V^T P^T  come out of ds_read_b64_tr_b16 (the hardware transposing read: a
//     16-lane group reads a [4 keys][16 d] block, lane i receives column i), two reads per 8-key MFMA operand;
//   * V tiles are double-buffered (issued a whole tile ahead), the K tile is re-filled behind the barrier that ends its QK phase and lands under the
//     softmax + PV phase: two barriers per key tile, nothing between them but matrix / softmax work;
//   * the last key tile computes only the key fragments that exist (S = 1037: 1 of 4 in QK, 1 of 2 key blocks in PV);
//   * XCD-aware block order: the query blocks of one (image, head) run on one XCD back to back, so its K / V rows are fetched from HBM once.
//
// LDS image (72 KiB per block, two blocks per CU):
//   Ks[3 planes][64 keys][128 B]   16-byte chunk c of row key at chunk c ^ (key & 7)                 (ds_read_b128 fragments: conflict-free)
//   Vs[2][3 planes][64 keys][128 B] 16-byte chunk c of row key at chunk c ^ (((key >> 1) & 3) << 1)   (tr reads of 8 consecutive rows x 32 B per
//                                   half-wave: the four even rows and the four odd rows each cover the four 32-byte blocks of their 128-byte half)
// Fragments:  S^T fragment kf (16 keys): A = K rows (lane (r = key, g): d = 32 kh + 8 g ..), B = Q (lane (r = query, g): same d) -> lane (query r, g)
// holds keys 16 kf + 4 g + e.  P operand of key block kk (32 keys): a lane's eight values are keys 32 kk + 4 g + e and 32 kk + 16 + 4 g + e (two
// accumulator quads); V^T fragment (16 d): lane (r = d, g) needs the same eight keys of column 16 fd + r = two tr reads of keys 32 kk + 16 a + 4 g ..+3.
#include "pf_common.h"
#include "../../include/pf_hip.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
typedef __attribute__((address_space(3))) v4s lds_v4s;

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// one 1-KiB LDS-DMA piece: lane L fetches 16 bytes at sbase + voff and the hardware writes them at lds_dst + 16 L.  (s_nop 4: the scalar base and M0 may
// come straight out of SALU instructions; hipcc pads nothing inside an asm statement.)
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void barrier_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
  const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)p0);
  const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)p1);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr int KS_BYTES = 3 * 64 * 128;          // one K (or V) tile: three planes x 64 keys x 128 bytes = 24 KiB = 24 DMA pieces

template <int QG>
__global__ __launch_bounds__(256, 2) void vit_attention_split3_v2_kernel(const bf16_t* __restrict__ qkv3, long plane_in, bf16_t* __restrict__ out3,
                                                                         long plane_out, int S, int Hh, float qscale, long kmaj_rows, int n_qb, int dbg) {
#ifdef PF_ATTN_DBG   // timing decomposition (results wrong by construction; make attndbg, tools/attn_split3_time.py): PF_ATTN_DBG bit 0 = no DMA inside the tile
#define ADBG(bit) (dbg & (bit))   // loop, 1 = no softmax VALU work, 2 = no QK MFMAs, 3 = no PV reads / MFMAs, 4 = no barriers
#else
#define ADBG(bit) 0
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ks = smem;
  char* const Vs = smem + KS_BYTES;                      // two stages
  constexpr int QW = 16 * QG, QB = 4 * QW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bh = bid / n_qb, qb = bid - bh * n_qb;
  const int b = bh / Hh, h = bh - b * Hh;
  const int q0 = qb * QB + wave * QW;
  const bool has_q = q0 < S;
  const int D = Hh * 64;
  const long rs = 3L * D;
  const bf16_t* base = qkv3 + (long)b * S * rs + h * 64;            // + plane * plane_in + s * rs (+ D for K, + 2 D for V)
  const int ntiles = (S + 63) / 64;

  // ---- DMA roles: wave w moves pieces 6 w .. 6 w + 5 of a tile (piece p = plane p >> 3, keys 8 (p & 7) .. + 7); lane L = row L >> 3, LDS chunk L & 7
  const unsigned rs_b = (unsigned)(rs * 2);                             // bytes between consecutive keys
  const unsigned krow_off = (unsigned)(lane >> 3) * rs_b + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
  const unsigned vrow_off = (unsigned)(lane >> 3) * rs_b + (unsigned)(((lane & 7) ^ (((lane >> 4) & 3) << 1)) << 4);
  const unsigned ks_lds = lds_addr_of(Ks), vs_lds = lds_addr_of(Vs);
  auto issue = [&](int kt, int which, unsigned lds_dst) {              // which = 1: K rows, 2: V rows
    const unsigned lane_off = (which == 1 ? krow_off : vrow_off) + (unsigned)(kt * 64) * rs_b;
    const char* sb = reinterpret_cast<const char*>(base + which * D);
    if (kt * 64 + 64 <= S) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int p = wave * 6 + i;
        dma16(sb + (long)(p >> 3) * plane_in * 2, lane_off + (unsigned)((p & 7) * 8) * rs_b, lds_dst + p * 1024);
      }
    } else {                                                            // keys past the sequence: re-read row S - 1 (finite values; their scores are masked)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int p = wave * 6 + i;
        const int over = max(kt * 64 + (p & 7) * 8 + (lane >> 3) - (S - 1), 0);
        dma16(sb + (long)(p >> 3) * plane_in * 2, lane_off + (unsigned)((p & 7) * 8 - over) * rs_b, lds_dst + p * 1024);
      }
    }
  };

  issue(0, 1, ks_lds);
  issue(0, 2, vs_lds);

  uint4 qf[QG][3][2];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const int qi = min(q0 + 16 * qg + r, S - 1);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) qf[qg][pl][kh] = *reinterpret_cast<const uint4*>(base + pl * plane_in + (long)qi * rs + 32 * kh + 8 * g);
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

  // per-lane LDS read bases (everything else is an immediate offset)
  //   K fragment (plane pl, fragment kf, half kh): row 16 kf + r, chunk (4 kh + g) ^ (r & 7)
  const char* kb0 = Ks + r * 128 + ((g ^ (r & 7)) << 4);
  const char* kb1 = Ks + r * 128 + (((4 + g) ^ (r & 7)) << 4);
  //   V tr read (plane pl, key block kk, half a, d fragment fd): row 32 kk + 16 a + 4 g + (r >> 2), 32-byte block fd ^ X, X = ((4 g + (r >> 2)) >> 1) & 3
  const int vrow = 4 * g + (r >> 2), vx = (vrow >> 1) & 3;
  const char* vb[4];
#pragma unroll
  for (int fd = 0; fd < 4; ++fd) vb[fd] = Vs + vrow * 128 + ((fd ^ vx) << 5) + ((r & 3) << 3);

  __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0) as a BUILTIN: hipcc then knows the Q loads have landed and puts no
  barrier_lds();                                         // vmcnt wait into the tile loop (an asm wait is invisible to it).  K(0), V(0) landed and visible

  auto tile = [&](int kt, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
    const int rem = S - kt * 64;                         // LAST: 1 .. 64 keys exist
    const int nkf = LAST ? (rem + 15) >> 4 : 4;
    const int nkk = LAST ? (rem + 31) >> 5 : 2;
    const int vstage = (kt & 1) * KS_BYTES;
    if (!LAST && !ADBG(1)) issue(kt + 1, 2, vs_lds + ((kt + 1) & 1) * KS_BYTES);
    // ---- S^T = K . Q^T : four 16-key fragments, six terms x two 32-deep halves, QG independent accumulator chains
    f32x4 sacc[QG][4];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) sacc[qg][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_q && !ADBG(4)) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        if (LAST && kf >= nkf) break;
        uint4 kfr[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          kfr[pl][0] = *reinterpret_cast<const uint4*>(kb0 + pl * 8192 + kf * 2048);
          kfr[pl][1] = *reinterpret_cast<const uint4*>(kb1 + pl * 8192 + kf * 2048);
        }
#define AT_TERM(PK, PQ)                                                                                                                   \
  _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int qg = 0; qg < QG; ++qg)                                      \
      sacc[qg][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kfr[PK][kh]), __builtin_bit_cast(bf16x8, qf[qg][PQ][kh]), \
                                                             sacc[qg][kf], 0, 0, 0);
        AT_TERM(0, 2) AT_TERM(2, 0) AT_TERM(1, 1) AT_TERM(0, 1) AT_TERM(1, 0) AT_TERM(0, 0)
#undef AT_TERM
      }
    }
    if (!ADBG(16)) barrier_lds();                        // every wave has read K(kt): the K stage is free
    if (!LAST && !ADBG(1)) issue(kt + 1, 1, ks_lds);
    if (has_q && ADBG(2)) {                              // (decomposition build: P operands straight from the accumulator bits, no VALU work)
#pragma unroll
      for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint4 a = __builtin_bit_cast(uint4, sacc[qg][2 * kk]), c = __builtin_bit_cast(uint4, sacc[qg][2 * kk + 1]);
#pragma unroll
          for (int fd = 0; fd < 4; ++fd) {
            bf16x8 vfr[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
              for (int k2 = 0; k2 < 2; ++k2) {
                const char* p = vb[fd] + vstage + pl * 8192 + k2 * 4096;
                vfr[pl][k2] = tr_pair(p, p + 2048);
              }
            if (!ADBG(8)) {
#pragma unroll
              for (int t = 0; t < 6; ++t)
                o[qg][fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[t % 3][kk], __builtin_bit_cast(bf16x8, (t & 1) ? a : c), o[qg][fd], 0, 0, 0);
            }
          }
        }
    } else if (has_q) {
      // ---- online softmax on base-2 logits (lane (query r, g) holds keys 16 kf + 4 g + e)
      uint4 pf[QG][3][2];                                // P as three bf16 planes, key block kk: elements t < 4 = fragment 2 kk, t >= 4 = fragment 2 kk + 1
      float alpha[QG];
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) {
        float mx = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = sacc[qg][kf][e];             // raw q.k: the positive scale commutes with the max and is folded into the exponent's FMA below
            if (LAST && 16 * kf + 4 * g + e >= rem) v = -INFINITY;
            sacc[qg][kf][e] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[qg], mx * qscale);
        alpha[qg] = __builtin_amdgcn_exp2f(m_run[qg] - m_new);
        m_run[qg] = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float pv[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            pv[t] = __builtin_amdgcn_exp2f(fmaf(sacc[qg][2 * kk + (t >> 2)][t & 3], qscale, -m_new));
            psum += pv[t];
          }
          uint32_t hw[4], mw[4], lw[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) split3_pair(pv[2 * u], pv[2 * u + 1], hw[u], mw[u], lw[u]);
          pf[qg][0][kk] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
          pf[qg][1][kk] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
          pf[qg][2][kk] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
        l_run[qg] = l_run[qg] * alpha[qg] + psum;
      }
      // ---- O^T = alpha O^T + V^T . P^T : four 16-row d fragments, two 32-key blocks, six terms
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) {
        if (ADBG(8)) { o[0][fd][0] += __uint_as_float(pf[0][0][0].x ^ pf[QG - 1][2][1].w); continue; }
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
          o[qg][fd][0] *= alpha[qg]; o[qg][fd][1] *= alpha[qg]; o[qg][fd][2] *= alpha[qg]; o[qg][fd][3] *= alpha[qg];
        }
        bf16x8 vfr[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            if (LAST && kk >= nkk) { vfr[pl][kk] = vfr[pl][0]; continue; }
            const char* p = vb[fd] + vstage + pl * 8192 + kk * 4096;
            vfr[pl][kk] = tr_pair(p, p + 2048);
          }
#define AT_TERM(PV, PP)                                                                                                                   \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                                                     \
    if (LAST && kk >= nkk) break;                                                                                                         \
    _Pragma("unroll") for (int qg = 0; qg < QG; ++qg) o[qg][fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                               \
        vfr[PV][kk], __builtin_bit_cast(bf16x8, pf[qg][PP][kk]), o[qg][fd], 0, 0, 0);                                                     \
  }
        AT_TERM(0, 2) AT_TERM(2, 0) AT_TERM(1, 1) AT_TERM(0, 1) AT_TERM(1, 0) AT_TERM(0, 0)
#undef AT_TERM
      }
    }
    if (!ADBG(16)) barrier_all();                        // every wave has read V(kt); K(kt + 1) and V(kt + 1) landed and visible
  };
  for (int kt = 0; kt + 1 < ntiles; ++kt) tile(kt, std::false_type{});
  tile(ntiles - 1, std::true_type{});

  if (!has_q) return;
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    float l = l_run[qg];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qo = q0 + 16 * qg + r;
    if (qo < S) {
      const long at = ((long)b * S + qo) * D + h * 64 + g * 4;
#pragma unroll
      for (int fd = 0; fd < 4; ++fd) {
        const float w4[4] = {o[qg][fd][0] * inv, o[qg][fd][1] * inv, o[qg][fd][2] * inv, o[qg][fd][3] * inv};
        store_split3(out3 + (kmaj_rows ? split3_at((long)b * S + qo, h * 64 + g * 4 + fd * 16, D, kmaj_rows) : at + fd * 16), plane_out, w4);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The PIPELINED kernel (round 6, the default).  Two measurements shaped it (profiles/r6_attention_v2.md, profiles/r6_issue_probe.md):
//  (1) the parts of the two-phase kernel above are ADDITIVE -- QK 68 us + PV 57 us (the matrix pipe's floor) + softmax 44 + DMA issue 22 + the rest: a
//      wave runs QK -> softmax -> PV one after the other, the partner wave on its SIMD covers only part of it (one block per CU 301 us, two 219 us;
//      MFMA pipe busy 42 %), and the 13-query tail blocks of S = 1037 run a whole third round alone;
//  (2) a wave cannot hide VALU work behind v_mfma_f32_16x16x32_bf16 AT ALL: {MFMA + K v_fma} costs 18.6 + 3.5 K cycles (the 4-pass instruction holds the
//      wave's issue for its whole 16 cycles), while v_mfma_f32_32x32x16_bf16 (8 passes, 32 cycles) hides six of them: 32.6 -> 38.6 cycles.
// So this kernel (a) multiplies on the 32 x 32 x 16 instruction -- one wave = 32 queries = the 32 columns of every accumulator tile -- and (b) overlaps the
// three stages inside every wave, on key BLOCKS of 32: in step j it issues the MFMAs of QK(j + 1) and PV(j - 1) and, between them, the float32 softmax
// of block j.  S^T and P live in two register sets each (named, parity = template parameter).
//   S^T = K . Q^T   tile [32 keys][32 queries]: A = K rows (lane (c = key, hi): d = 16 ks + 8 hi ..), B = Q (lane (c = query, hi): the same d), four 16-deep
//                   steps ks x six plane pairs; lane (query c, hi) holds keys (reg & 3) + 8 (reg >> 2) + 4 hi.
//   O^T = V^T . P^T tiles [32 d][32 queries], t = 0, 1: two 16-key steps s; a lane's eight P values of step s are accumulator registers 8 s .. 8 s + 7 =
//                   keys 16 s + 4 hi + e and 16 s + 8 + 4 hi + e -- the V^T operand (lane (c = d, hi)) is two transposing reads of those key quads.
//   An MFMA "super-group" n = 0..3 = the six QK products of step ks = n on the S chain interleaved with the six PV products of (s = n >> 1, t = n & 1) on
//   the O[t] chain (consecutive MFMAs never share an accumulator); its nine LDS reads are issued one super-group ahead (two fragment register sets); the
//   softmax is cut into four slices that ride between the MFMAs (sched_group_barrier: MFMA, then up to six others); its head (row maximum, new running
//   maximum, rescale factor) covers the first reads' latency.  O is rescaled only in steps where some row's running maximum moved (wave-uniform
//   branch; the factor is exactly 1.0 otherwise, so skipping is exact).
//   K ring: 3 slots x [3 planes][32 keys][128 B] = 36 KiB, V ring the same: 72 KiB per block, two blocks per CU.
//     K rows: 16-byte chunk c of row key at chunk c ^ ((key >> 1) & 7)   (the 16-lane groups of ds_read_b128 are rows {0-3, 12-15, 20-27} / {4-11, 16-19,
//             28-31}: eight distinct (key >> 1) & 7 per row parity)
//     V rows: 32-byte block q of row key at block q ^ (((key >> 1) & 1) << 1)   (a half-wave's transposing reads cover four consecutive rows x two blocks)
//   step j reads K slot (j + 1) % 3 and V slot (j - 1) % 3; ONE barrier ends it (s_waitcnt vmcnt(6): the six pieces issued two steps ago -- K(j + 2),
//   V(j) -- have landed), then K(j + 4) and V(j + 2) are issued into the two slots just released: every block has two whole steps to land.
//   Blocks past the sequence are fetched as clamped re-reads of row S - 1 (uniform vmcnt accounting) and never enter a softmax.
//   Grid order: the 13-query tail blocks of every (image, head) come FIRST, so they run beside full blocks instead of alone at the end.
// Arithmetic: the same six products, smallest first, per accumulator; the online softmax advances in 32-key blocks, so results agree with version 1 to
// float32 rounding, not bit for bit (tests/op_checks.py vit_attention_split3_v2: error against float64 <= version 1's).
typedef __attribute__((ext_vector_type(16))) float f32x16;
#ifdef PF_ATTN_DBG
// s_memtime timeline of the pipelined kernel (decomposition build only): per-phase cycle sums of wave 0 of three blocks, read back by pf_attn_dbg_timeline
__device__ long long g_attn_tl[3][16];
// ... and the life of EVERY block: {s_memrealtime at entry, at exit (100 MHz, chip-wide), HW_ID, XCC_ID} of its wave 0 (pf_attn_dbg_blocks)
__device__ long long g_attn_blk[4096][4];
#define TL_DECL long long tl_prev = 0, tl_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const int tl_blk = blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 1 : blockIdx.x == gridDim.x - 1 ? 2 : -1; const bool tl_on = tl_blk >= 0 && wave == 0
#define TL_START() do { tl_prev = __builtin_amdgcn_s_memtime(); } while (0)
// (branch-free: every wave stamps, so that the marks do not split the scheduling regions; scalar instructions only)
#define TL_MARK(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[k] += t_ - tl_prev; tl_prev = t_; } while (0)
#define TL_FLUSH() do { if (tl_on && lane == 0) { for (int k_ = 0; k_ < 12; ++k_) g_attn_tl[tl_blk][k_] = tl_acc[k_]; } } while (0)
#else
#define TL_DECL
#define TL_START()
#define TL_MARK(k)
#define TL_FLUSH()
#endif
constexpr int PB_BYTES = 3 * 32 * 128;          // one K (or V) block: three planes x 32 keys x 128 bytes = 12 KiB = 12 DMA pieces
constexpr int PIPE_LDS = 6 * PB_BYTES;
constexpr float RESCALE_THR = 24.0f;      // binary orders a row's block maximum may exceed its exponent reference before O and l are rescaled

// v_cvt_pk_bf16_f32 as the COMPILER's instruction (a vector conversion it selects itself): unlike the asm form of pf_common.h it takes part in instruction
// scheduling (sched_group_barrier only places instructions the scheduler can classify).  Same instruction, same round-to-nearest-even result.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t cvt_pk_sched(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ void split3_pair_sched(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
#pragma clang fp contract(off)
  h = cvt_pk_sched(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_sched(ra, rb);
  l = cvt_pk_sched(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));      // (fmaxf would add a canonicalising v_max per MFMA output)
  return r;
}

__global__ __launch_bounds__(256, 2) void vit_attention_split3_pipe_kernel(const bf16_t* __restrict__ qkv3, long plane_in, bf16_t* __restrict__ out3,
                                                                           long plane_out, int S, int Hh, float qscale, long kmaj_rows, int n_qb, int BH, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Kr = smem;
  char* const Vr = smem + 3 * PB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, hi = lane >> 5;
  // block order: when every (image, head) ends in a partial query block, those BH tail blocks take the first BH ids (b % 8 = the XCD they run on, and
  // the full blocks of (image, head) bh follow on XCD bh % 8 too when BH % 8 == 0); otherwise the XCD-contiguous order of the two-phase kernel
  int bh, qb;
  {
    const int bid = (int)blockIdx.x;
    const bool tail = (S & 127) != 0 && (BH & 7) == 0 && n_qb > 1;
    if (tail) {
      if (bid < BH) { bh = bid; qb = n_qb - 1; }
      else {
        const int l = bid - BH, idx = l >> 3, nf = n_qb - 1;
        bh = (idx / nf) * 8 + (l & 7);
        qb = idx % nf;
      }
    } else {
      const int l = xcd_remap(bid, (int)gridDim.x);
      bh = l / n_qb;
      qb = l - bh * n_qb;
    }
  }
  const int b = bh / Hh, h = bh - b * Hh;
  const int NB = (S + 31) / 32;
  // KEY-SPLIT tail blocks: the last query block of S = 1037 holds 13 queries -- one wave's worth.  Run as it stands that block takes as long as a full one (a
  // wave's chain over all 33 key blocks: 67 of 79 us, profiles/r6_attention_blocks.md) with three waves idle, and the 128 of them cost the launch a third
  // round of blocks.  Instead all four waves take the SAME (at most 32) queries and wave w the key blocks jb = w (mod 4): in step j it runs the one stage
  // whose block is its own (QK(j + 1), softmax(j) or PV(j - 1)), so a step of the block costs one stage instead of three; the four partial (reference
  // exponent, sum, O) sets are merged through LDS at the end (flash-decoding style: O = sum_w O_w 2^(m_w - m), l likewise).
  const bool ksplit = S - qb * 128 <= 32 && NB >= 8;
  const int q0 = qb * 128 + (ksplit ? 0 : wave * 32);
  const bool has_q = q0 < S;
  const int D = Hh * 64;
  const long rs = 3L * D;
  const bf16_t* base = qkv3 + (long)b * S * rs + h * 64;

  // ---- DMA roles: wave w moves pieces 3 w .. 3 w + 2 of a block (piece p = plane p >> 2, keys 8 (p & 3) .. + 7); lane L = row L >> 3, LDS chunk L & 7
  const unsigned rs_b = (unsigned)(rs * 2);
  const unsigned kchunk = (unsigned)(((lane & 7) ^ ((lane >> 4) & 3)) << 4);        // ^ 64 for the odd pieces: (key >> 1) & 7 = 4 (p & 1) + (L >> 4)
  const unsigned vchunk = (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 2)) << 4);
  const unsigned kr_lds = lds_addr_of(Kr), vr_lds = lds_addr_of(Vr);
  auto issue = [&](int jb, int which, unsigned lds_dst) {              // which = 1: K rows, 2: V rows of key block jb
    const char* sb = reinterpret_cast<const char*>(base + which * D);
    const unsigned chunk = which == 1 ? kchunk : vchunk;
    if (jb * 32 + 32 <= S) {
      const unsigned lane_off = (unsigned)(jb * 32 + (lane >> 3)) * rs_b;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int p = wave * 3 + i;
        const unsigned ch = which == 1 ? chunk ^ (unsigned)((p & 1) << 6) : chunk;
        dma16(sb + (long)(p >> 2) * plane_in * 2, lane_off + ch + (unsigned)((p & 3) * 8) * rs_b, lds_dst + p * 1024);
      }
    } else {                                                            // keys past the sequence: re-read row S - 1 (finite values; masked or never used)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int p = wave * 3 + i;
        const unsigned ch = which == 1 ? chunk ^ (unsigned)((p & 1) << 6) : chunk;
        const int row = min(jb * 32 + (p & 3) * 8 + (lane >> 3), S - 1);
        dma16(sb + (long)(p >> 2) * plane_in * 2, ch + (unsigned)row * rs_b, lds_dst + p * 1024);
      }
    }
  };
  issue(0, 1, kr_lds);
  issue(1, 1, kr_lds + PB_BYTES);
  issue(2, 1, kr_lds + 2 * PB_BYTES);
  issue(0, 2, vr_lds);

  uint4 qf[3][4];                                        // Q planes, four 16-deep steps: d = 16 ks + 8 hi ..
  {
    const int qi = min(q0 + c, S - 1);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[pl][ks] = *reinterpret_cast<const uint4*>(base + pl * plane_in + (long)qi * rs + 16 * ks + 8 * hi);
  }
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f, a_pend = 1.f;
  f32x16 sA, sB;                                         // S^T of the even / odd key blocks
  uint4 pA[3][2], pB[3][2];                              // P planes of the even / odd key blocks: [plane][16-key step]
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int i = 0; i < 2; ++i) pA[pl][i] = pB[pl][i] = make_uint4(0, 0, 0, 0);

  // per-lane LDS read bases (slot, plane and step offsets are immediates / one scalar)
  const char* kbase[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kbase[ks] = Kr + c * 128 + (((2 * ks + hi) ^ ((c >> 1) & 7)) << 4);
  const int j16 = lane & 15, jb1 = (j16 >> 3) & 1;
  const char* vbase[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) vbase[t] = Vr + (4 * hi + (j16 >> 2)) * 128 + ((2 * (t ^ jb1) + (c >> 4)) << 5) + ((j16 & 3) << 3);

  TL_DECL;
#ifdef PF_ATTN_DBG
  const long long blk_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  TL_START();
  __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0) as a builtin (hipcc then knows the Q loads have landed)
  barrier_lds();
  TL_MARK(0);                                            // [0] prologue wait

  auto step = [&](auto qk_c, auto sm_c, auto mask_c, auto pv_c, auto par_c, int ks_off, int vs_off, int rem) {
    constexpr bool QK = decltype(qk_c)::value, SM = decltype(sm_c)::value, MASK = decltype(mask_c)::value, PV = decltype(pv_c)::value;
    constexpr int PAR = decltype(par_c)::value;
    f32x16& s_next = PAR ? sA : sB;
    f32x16& s_cur = PAR ? sB : sA;
    uint4 (&p_cur)[3][2] = PAR ? pB : pA;
    uint4 (&p_prev)[3][2] = PAR ? pA : pB;
    if (!has_q) return;
    uint4 kfs[2][3];                                     // [register set][plane]
    bf16x8 vfs[2][3];
    float pv[16];
    float m_new = 0.f;
    auto load_group = [&](auto n_c) {
      constexpr int n = decltype(n_c)::value;
      if constexpr (QK) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) kfs[n & 1][pl] = *reinterpret_cast<const uint4*>(kbase[n] + ks_off + pl * 4096);
      }
      if constexpr (PV) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const char* p = vbase[n & 1] + vs_off + pl * 4096 + (n >> 1) * 2048;
          vfs[n & 1][pl] = tr_pair(p, p + 1024);
        }
      }
    };
    // one MFMA slot of super-group n: slot i = 0..11; even slots multiply on the S chain, odd slots on the O[n & 1] chain (term i >> 1 of the six plane
    // pairs, smallest first: (0, 2) (2, 0) (1, 1) (0, 1) (1, 0) (0, 0)); a variant without QK (or PV) has six slots
    auto mfma_slot = [&](auto n_c, auto i_c) {
      constexpr int n = decltype(n_c)::value, i = decltype(i_c)::value;
      constexpr int term = (QK && PV) ? (i >> 1) : i;
      constexpr bool on_s = QK && (!PV || (i & 1) == 0);
      constexpr int PA = term == 0 ? 0 : term == 1 ? 2 : term == 2 ? 1 : term == 3 ? 0 : term == 4 ? 1 : 0;
      constexpr int PB_ = term == 0 ? 2 : term == 1 ? 0 : term == 2 ? 1 : term == 3 ? 1 : term == 4 ? 0 : 0;
      if constexpr (on_s) {
        if constexpr (n == 0 && term == 0)
          s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfs[n & 1][PA]), __builtin_bit_cast(bf16x8, qf[PB_][n]),
                                                           f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else
          s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfs[n & 1][PA]), __builtin_bit_cast(bf16x8, qf[PB_][n]), s_next, 0, 0, 0);
      } else {
        o[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfs[n & 1][PA], __builtin_bit_cast(bf16x8, p_prev[PB_][n >> 1]), o[n & 1], 0, 0, 0);
      }
    };
    // ---- softmax slices (float32, base-2 logits; lane (query c, hi) holds keys (reg & 3) + 8 (reg >> 2) + 4 hi of block j)
    auto sm_head = [&]() {
      if (MASK) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if ((i & 3) + 8 * (i >> 2) + 4 * hi >= rem) s_cur[i] = -INFINITY;
      }
      // row maximum: a three-level tree of v_max3 (the eight-deep chain it replaces sat in front of every step), then lane c + 32's half of the keys
      const float t0 = max3f(s_cur[0], s_cur[1], s_cur[2]), t1 = max3f(s_cur[3], s_cur[4], s_cur[5]), t2 = max3f(s_cur[6], s_cur[7], s_cur[8]);
      const float t3 = max3f(s_cur[9], s_cur[10], s_cur[11]), t4 = max3f(s_cur[12], s_cur[13], s_cur[14]);
      const float u0 = max3f(t0, t1, t2), u1 = max3f(t3, t4, s_cur[15]);
      float mx = max3f(u0, u1, u1);
      const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = max3f(__uint_as_float(y[0]), __uint_as_float(y[1]), __uint_as_float(y[1]));
      // DEFERRED rescale: the exponent reference m_run of a row moves only when a block's maximum exceeds it by more than RESCALE_THR binary orders (always
      // on the first block: m_run = -inf).  P = 2^(s - m_run) may then exceed 1 -- by at most 2^THR; float32 (and its three-plane split) is scale-free, so
      // nothing is lost, and O / l carry the same factor, which cancels in O / l.  With the textbook rule (reference = running maximum) some row of the 32
      // moves in most blocks (P(any of 32 rows beats its maximum over j earlier blocks) = 1 - (j / (j + 1))^32) and the 32-register rescale sat in front
      // of nearly every step; now the factor is exactly 1 after the first block unless the logits grow by 2^THR, and the rescale is a cold branch.
      const float mb = mx * qscale;
      const bool need = mb > m_run + RESCALE_THR;
      m_new = need ? mb : m_run;
      a_pend = need ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.0f;
      m_run = m_new;
    };
    float psum = 0.f, ra[8], rb[8];
    auto sm_exp1 = [&](int i) {
      pv[i] = __builtin_amdgcn_exp2f(fmaf(s_cur[i], qscale, -m_new));
      psum += pv[i];
    };
    // pair u of the sixteen probabilities -> word u & 3 of step u >> 2 of the three planes, in two halves (x = h + m + l: csrc/pf_common.h split3_pair)
    auto sm_split_a = [&](int u) {
#pragma clang fp contract(off)
      const uint32_t hw = cvt_pk_sched(pv[2 * u], pv[2 * u + 1]);
      ra[u] = pv[2 * u] - __uint_as_float(hw << 16);
      rb[u] = pv[2 * u + 1] - __uint_as_float(hw & 0xffff0000u);
      (&p_cur[0][u >> 2].x)[u & 3] = hw;
    };
    auto sm_split_b = [&](int u) {
#pragma clang fp contract(off)
      const uint32_t mw = cvt_pk_sched(ra[u], rb[u]);
      const uint32_t lw = cvt_pk_sched(ra[u] - __uint_as_float(mw << 16), rb[u] - __uint_as_float(mw & 0xffff0000u));
      (&p_cur[1][u >> 2].x)[u & 3] = mw;
      (&p_cur[2][u >> 2].x)[u & 3] = lw;
    };
    constexpr int NSLOT = (QK ? 6 : 0) + (PV ? 6 : 0);                // MFMA slots per super-group
    // the softmax work that rides behind slot i of super-group n (every slot ends in a scheduling fence: the order below IS the instruction order)
    auto valu_chunk = [&](auto n_c, auto i_c) {
      constexpr int n = decltype(n_c)::value, i = decltype(i_c)::value;
      if constexpr (SM) {
        if constexpr (NSLOT == 12) {
          if constexpr (n == 0) {                          // sixteen exponentials over twelve slots
            if constexpr (i < 8) sm_exp1(i);
            else { sm_exp1(8 + 2 * (i - 8)); sm_exp1(9 + 2 * (i - 8)); }
            if constexpr (i == 11) l_run = l_run * a_pend + psum;
          } else {                                         // sixteen half-splits over the even slots of the remaining 36
            constexpr int gs = 12 * (n - 1) + i;
            if constexpr ((gs & 1) == 0 && gs / 2 < 16) {
              if constexpr (((gs / 2) & 1) == 0) sm_split_a(gs / 4);
              else sm_split_b(gs / 4);
            }
          }
        } else if constexpr (NSLOT == 6) {                 // (first / last steps: six slots per super-group)
          if constexpr (n == 0) {
            sm_exp1(2 * i); sm_exp1(2 * i + 1);
            if constexpr (i < 2) { sm_exp1(12 + 2 * i); sm_exp1(13 + 2 * i); }
            if constexpr (i == 5) l_run = l_run * a_pend + psum;
          } else {
            constexpr int gs = 6 * (n - 1) + i;
            if constexpr (gs < 16) {
              if constexpr ((gs & 1) == 0) sm_split_a(gs / 2);
              else sm_split_b(gs / 2);
            }
          }
        }
      }
    };
    if constexpr (PV) {
      // O is relative to the running maximum of two blocks ago: rescale by the factor the previous step's softmax left -- only if some row's maximum moved
      const float a_use = a_pend;
      if (__builtin_amdgcn_ballot_w64(a_use != 1.0f) != 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[t][i] *= a_use;
      }
    }
    if constexpr (QK || PV) load_group(std::integral_constant<int, 0>{});
    if constexpr (SM) sm_head();
    __builtin_amdgcn_sched_barrier(0);
    TL_MARK(4);                                          // [4] rescale test + first reads + softmax head
    auto slot = [&](auto n_c, auto i_c) {
      constexpr int i = decltype(i_c)::value;
      if constexpr (i < NSLOT) {
        mfma_slot(n_c, i_c);
        valu_chunk(n_c, i_c);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto group = [&](auto n_c) {
      constexpr int n = decltype(n_c)::value;
      if constexpr (NSLOT > 0) {
        if constexpr (n < 3) {
          load_group(std::integral_constant<int, n + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        slot(n_c, std::integral_constant<int, 0>{});
        slot(n_c, std::integral_constant<int, 1>{});
        slot(n_c, std::integral_constant<int, 2>{});
        slot(n_c, std::integral_constant<int, 3>{});
        slot(n_c, std::integral_constant<int, 4>{});
        slot(n_c, std::integral_constant<int, 5>{});
        slot(n_c, std::integral_constant<int, 6>{});
        slot(n_c, std::integral_constant<int, 7>{});
        slot(n_c, std::integral_constant<int, 8>{});
        slot(n_c, std::integral_constant<int, 9>{});
        slot(n_c, std::integral_constant<int, 10>{});
        slot(n_c, std::integral_constant<int, 11>{});
      }
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
    if constexpr (NSLOT == 0 && SM) {                    // (a softmax alone: the single-key-block sequence)
#pragma unroll
      for (int i = 0; i < 16; ++i) sm_exp1(i);
      l_run = l_run * a_pend + psum;
#pragma unroll
      for (int u = 0; u < 8; ++u) { sm_split_a(u); sm_split_b(u); }
    }
  };
  // end of a step: the pieces issued two steps ago have landed (at most the six of the previous step stay in flight), every wave is done with the
  // two slots the step read; then refill them
  auto turn = [&](int j, int ks, int vs) {
    TL_MARK(8);                                          // [8] the four super-groups
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TL_MARK(1);                                          // [1] landing wait + barrier
    issue(j + 4, 1, kr_lds + ks);
    issue(j + 2, 2, vr_lds + vs);
    TL_MARK(2);                                          // [2] DMA issue
  };
  using T = std::true_type;
  using F = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  if (!ksplit) {
    // slot byte offsets: K slot of step j = ((j + 1) % 3) PB, V slot = ((j + 2) % 3) PB
    step(T{}, F{}, F{}, F{}, P1{}, 0, 0, 0);                                           // j = -1: QK(0)
    turn(-1, 0, PB_BYTES);
    if (NB > 1) step(T{}, T{}, F{}, F{}, P0{}, PB_BYTES, 0, 0);                        // j = 0: QK(1), softmax(0)
    else step(F{}, T{}, T{}, F{}, P0{}, 0, 0, S);                                      //        (a single key block: softmax(0) masked)
    turn(0, PB_BYTES, 2 * PB_BYTES);
    int ks = 2 * PB_BYTES, vs = 0;                                                     // step 1: K slot 2, V slot 0
    int j = 1;
    for (; j + 1 <= NB - 2; j += 2) {
      step(T{}, T{}, F{}, T{}, P1{}, ks, vs, 0);
      turn(j, ks, vs);
      ks = ks == 2 * PB_BYTES ? 0 : ks + PB_BYTES;
      vs = vs == 2 * PB_BYTES ? 0 : vs + PB_BYTES;
      step(T{}, T{}, F{}, T{}, P0{}, ks, vs, 0);
      turn(j + 1, ks, vs);
      ks = ks == 2 * PB_BYTES ? 0 : ks + PB_BYTES;
      vs = vs == 2 * PB_BYTES ? 0 : vs + PB_BYTES;
    }
    if (j <= NB - 2) {                                                                 // one more steady step (odd j)
      step(T{}, T{}, F{}, T{}, P1{}, ks, vs, 0);
      turn(j, ks, vs);
      ks = ks == 2 * PB_BYTES ? 0 : ks + PB_BYTES;
      vs = vs == 2 * PB_BYTES ? 0 : vs + PB_BYTES;
      ++j;
    }
    // j == max(NB - 1, 1): the last block's softmax (masked) beside PV(NB - 2), then PV(NB - 1); nothing is fetched any more
    const int rem = S - (NB - 1) * 32;
    if (NB > 1) {
      if (j & 1) step(F{}, T{}, T{}, T{}, P1{}, ks, vs, rem);
      else step(F{}, T{}, T{}, T{}, P0{}, ks, vs, rem);
      vs = vs == 2 * PB_BYTES ? 0 : vs + PB_BYTES;
      ++j;
    }
    barrier_all();                                          // V(NB - 1) of EVERY wave has landed (and every DMA, before the block may exit)
    if (j & 1) step(F{}, F{}, F{}, T{}, P1{}, 0, vs, 0);
    else step(F{}, F{}, F{}, T{}, P0{}, 0, vs, 0);
  } else {
    // one stage per wave and step (see ksplit above); the ring protocol (turn) is the same for every wave
    const int remk = S - (NB - 1) * 32;
    int ksj = 0, vsj = PB_BYTES;                                                       // j = -1: K slot 0, V slot 1
    // (a wave's stages of one block are three consecutive steps and its next block is four steps later: ONE S^T and ONE P register set suffice -- QK writes
    //  sA (parity 1), the softmax reads sA and writes pA (parity 0), PV reads pA (parity 1) -- so the only instantiation this path adds is the bare softmax)
    for (int jj = -1; jj <= NB; ++jj) {
      if (((jj + 1 - wave) & 3) == 0 && jj + 1 < NB) step(T{}, F{}, F{}, F{}, P1{}, ksj, vsj, 0);            // QK of my block jj + 1
      else if (((jj - wave) & 3) == 0 && jj >= 0 && jj < NB) {                                                // softmax of my block jj
        if (jj == NB - 1) step(F{}, T{}, T{}, F{}, P0{}, ksj, vsj, remk);
        else step(F{}, T{}, F{}, F{}, P0{}, ksj, vsj, 0);
      } else if (((jj - 1 - wave) & 3) == 0 && jj >= 1 && jj - 1 < NB) step(F{}, F{}, F{}, T{}, P1{}, ksj, vsj, 0);   // PV of my block jj - 1
      if (jj <= NB - 2) turn(jj, ksj, vsj);
      else if (jj == NB - 1) barrier_all();
      ksj = ksj == 2 * PB_BYTES ? 0 : ksj + PB_BYTES;
      vsj = vsj == 2 * PB_BYTES ? 0 : vsj + PB_BYTES;
    }
    // merge: waves 1..3 park (reference exponent, sum, O) in LDS (the rings are idle: every DMA landed before the last barrier), wave 0 folds them in
    __syncthreads();
    float* park = reinterpret_cast<float*>(smem);
    if (wave > 0) {
      float* pw = park + (wave - 1) * 34 * 64 + lane;
      pw[0] = m_run;
      pw[64] = l_run;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) pw[(2 + t * 16 + i) * 64] = o[t][i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int w = 0; w < 3; ++w) {
      const float* pw = park + w * 34 * 64 + lane;
      const float mw = pw[0], lw = pw[64];
      const float mn = fmaxf(m_run, mw);
      const float f0 = __builtin_amdgcn_exp2f(m_run - mn), fw = __builtin_amdgcn_exp2f(mw - mn);      // (a wave without key blocks parks -inf, 0, 0: fw = 0)
      l_run = l_run * f0 + lw * fw;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = o[t][i] * f0 + pw[(2 + t * 16 + i) * 64] * fw;
      m_run = mn;
    }
  }

  TL_MARK(10);                                           // [10] final wait
  if (!has_q) return;
  float l = l_run;
  {
    const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
    l = __uint_as_float(y[0]) + __uint_as_float(y[1]);
  }
  const float inv = 1.0f / l;
  const int qo = q0 + c;
  if (qo < S) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int col = h * 64 + 32 * t + 8 * rq + 4 * hi;
        const float w4[4] = {o[t][4 * rq] * inv, o[t][4 * rq + 1] * inv, o[t][4 * rq + 2] * inv, o[t][4 * rq + 3] * inv};
        store_split3(out3 + split3_at((long)b * S + qo, col, D, kmaj_rows), plane_out, w4);
      }
  }
  TL_MARK(11);                                           // [11] epilogue
  TL_FLUSH();
#ifdef PF_ATTN_DBG
  if (tid == 0 && blockIdx.x < 4096) {
    g_attn_blk[blockIdx.x][0] = blk_t0;
    g_attn_blk[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    g_attn_blk[blockIdx.x][2] = __builtin_amdgcn_s_getreg((31 << 11) | 4)   /* HW_REG_HW_ID */;
    g_attn_blk[blockIdx.x][3] = __builtin_amdgcn_s_getreg((31 << 11) | 20)   /* HW_REG_XCC_ID */;
  }
#endif
}

constexpr int V2_LDS = 3 * KS_BYTES;

}  // namespace

// C entry: see include/pf_hip.h.  schedule = 1: the two-phase kernel (queries_per_wave = 16 or 32, 0 = 32 when the grid of 128-query blocks fills the chip,
// else 16), 2: the pipelined kernel (always 32 queries per wave), 0 = default (pipelined).
extern "C" int pf_vit_attention_split3_v2(const void* qkv3, long plane_in, void* out3, long plane_out, int kmajor, int B, int S, int Hh,
                                          int queries_per_wave, int schedule, void* stream) {
  if (!qkv3 || !out3 || B <= 0 || S <= 0 || Hh <= 0 || plane_in < (long)B * S * Hh * 192 || plane_out < (long)B * S * Hh * 64) return PF_ERR_ARG;
  if ((long)S * Hh * 192 * 2 >= (1L << 31)) return PF_ERR_ARG;                           // 32-bit byte offsets inside one image's rows
  if (queries_per_wave != 0 && queries_per_wave != 16 && queries_per_wave != 32) return PF_ERR_ARG;
  if (schedule < 0 || schedule > 2) return PF_ERR_ARG;
  const float qscale = 0.125f * 1.4426950408889634f;        // head_dim^-1/2 (attention.py:55) times log2(e): base-2 softmax
  int dbg = 0, lds2 = V2_LDS, ldsp = PIPE_LDS;
#ifdef PF_ATTN_DBG
  if (const char* e = getenv("PF_ATTN_DBG")) dbg = atoi(e);
  if (const char* e = getenv("PF_ATTN_LDS")) lds2 = ldsp = atoi(e);      // (bytes of dynamic LDS requested: > 80 KiB leaves ONE block per CU)
#endif
  static int attr_done = 0;
  if (attr_done != lds2) {
    const void* ks[3] = {reinterpret_cast<const void*>(vit_attention_split3_v2_kernel<1>), reinterpret_cast<const void*>(vit_attention_split3_v2_kernel<2>),
                         reinterpret_cast<const void*>(vit_attention_split3_pipe_kernel)};
    for (int i = 0; i < 3; ++i)
      if (hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 2 ? lds2 : ldsp) != hipSuccess) return PF_ERR_LAUNCH;
    attr_done = lds2;
  }
  int qw = queries_per_wave;
  if (schedule != 1) qw = 32;                                 // the pipelined kernel: a wave = the 32 columns of its accumulator tiles
  if (qw == 0) qw = (long)((S + 127) / 128) * B * Hh >= 512 ? 32 : 16;
  const int n_qb = (S + 4 * qw - 1) / (4 * qw);
  const long kmaj_rows = kmajor ? (long)B * S : 0L;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(n_qb * B * Hh), block(256);
#define ATTN_LAUNCH(KERNEL, LDS)                                                                                                          \
  hipLaunchKernelGGL(KERNEL, grid, block, LDS, st, (const bf16_t*)qkv3, plane_in, (bf16_t*)out3, plane_out, S, Hh, qscale, kmaj_rows, n_qb, dbg)
  if (schedule == 1) {
    if (qw == 32) ATTN_LAUNCH(vit_attention_split3_v2_kernel<2>, lds2);
    else ATTN_LAUNCH(vit_attention_split3_v2_kernel<1>, lds2);
  } else {
    hipLaunchKernelGGL(vit_attention_split3_pipe_kernel, grid, block, ldsp, st, (const bf16_t*)qkv3, plane_in, (bf16_t*)out3, plane_out, S, Hh, qscale, kmaj_rows, n_qb,
                       B * Hh, dbg);
  }
#undef ATTN_LAUNCH
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

#ifdef PF_ATTN_DBG
// decomposition build only: the s_memtime sums of the last pipelined launch (3 blocks x 16 slots, host buffer of 48 long long)
extern "C" int pf_attn_dbg_blocks(long long* out, int nblocks) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_blk), sizeof(long long) * 4 * (nblocks < 4096 ? nblocks : 4096)) == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}
extern "C" int pf_attn_dbg_timeline(long long* out48) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_attn_tl), sizeof(long long) * 48) == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}
#endif
