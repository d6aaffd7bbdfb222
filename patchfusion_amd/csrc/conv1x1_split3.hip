// 1x1 convolution / linear layer with FLOAT32 activations on the bf16 matrix cores (round 6): the split-precision product of
// csrc/gemm_split3.hip -- x = x_h + x_m + x_l, w = w_h + w_m + w_l in bf16, the six leading partial products accumulated in float32, smallest
// first -- for layers whose producers write plain float32 NHWC tensors (the DPT / metric-bins heads' 1x1 convolutions, localbins_layers.py:99-117,
// attractor.py:156-161, the G2L Swin linears, swin_layers.py:120-128,133-164).  Until round 5 these ran on v_mfma_f32_16x16x4_f32 (1/16 of the bf16
// rate, 0.44-0.70 of THAT pipe); a stand-alone split pass in front of pf_gemm_split3 would move as many bytes as the f32 kernel loses, so the split
// happens HERE, in the loader: a thread reads 8 floats of a token row, splits them (33 VALU instructions) and writes one 16-byte slot of each of the
// three LDS planes; the weights arrive pre-split (packing.pack_conv_split3, chunk-major) by LDS-DMA.
//
//   D[n][m] = sum_k W[n][k] X[m][k]     MFMA A = weight rows, B = token rows; a lane's four accumulator registers = four consecutive channels
//                                        of one token (float4 stores along the channel axis), exactly as in gemm_split3.hip / igemm.hip
// Tile 64 (or 128) tokens x 128 channels, K chunks of 32, four (eight) waves of 32 x 64 (FM = 2, FN = 4: 48 MFMAs per chunk and wave), two LDS stages of
// 36 (48) KiB ([X h|m|l][W h|m|l], 64-byte rows, 16-byte slot g of row r at g ^ ((r >> 1) & 3): conflict-free ds_read_b128 fragments).
// Pipeline, ONE barrier per chunk: at the top of chunk kc (after the barrier that retires the reads of stage kc-1) the wave issues the DMA of
// W(kc+1) and writes the split of X(kc+1) -- read from HBM TWO chunks earlier, two register sets -- into the other stage, issues the global loads of
// X(kc+3), then reads its fragments of chunk kc and multiplies.  X is read exactly once per channel tile.
// Epilogue as pf_conv: (act(v + bias) * scale) + res + res2, float32 out.
#include <cstdlib>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned int c1_zero_page[64];

__device__ __forceinline__ unsigned c1_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void c1_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void c1_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void c1_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int BN = 128, WN = 2, WTM = 32, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
constexpr int WPLANE = BN * 64;                   // bytes of one 128-row weight plane of a stage

// eight floats -> one 16-byte slot of each of the three LDS planes of a stage
template <int XPLANE>
__device__ __forceinline__ void c1_xwrite(char* d, f32x4 xa, f32x4 xb, bool ok, int relu_in) {
  float v[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
  if (!ok) {                                    // token rows beyond M: the (clamped) load's values are not used
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  }
  if (relu_in) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
  *reinterpret_cast<uint4*>(d) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(d + XPLANE) = make_uint4(m[0], m[1], m[2], m[3]);
  *reinterpret_cast<uint4*>(d + 2 * XPLANE) = make_uint4(l[0], l[1], l[2], l[3]);
}

// BM = 64: 72 KiB of LDS, four waves -- TWO blocks per CU, so that one block's prologue (the first HBM round trip) and store tail run beside the
// other's MFMAs; BM = 128: 96 KiB, eight waves, one block per CU (measured slower on every layer of the pass: nothing covers fill and drain).
// (Measured and NOT kept, round 6, profiles/r6_conv1x1_split3.md: the weight pieces through one or two register sets instead of LDS-DMA -- two chunks
// of latency for W as well -- 0.89-0.95x; s_setprio around the MFMAs to de-phase the two blocks of a CU +-0; the 128-token tile 0.95x.)
template <int BM>
__global__ __launch_bounds__(4 * BM, 2) void conv1x1_split3_kernel(const pf_conv_params p, const bf16_t* __restrict__ w3, int w_rows, long w_bstride,
                                                                int mt, int nt) {
  constexpr int WM = BM / WTM, NW = WM * WN;
  constexpr int XPLANE = BM * 64;                  // bytes of one token plane of a stage
  constexpr int STAGE = 3 * XPLANE + 3 * WPLANE;   // X h | m | l | W h | m | l
  constexpr int WPW = 24 / NW;                     // DMA pieces (16 rows x 64 B) per wave and chunk: 3 planes x 8 pieces
  static_assert(24 % NW == 0, "weight pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = p.B * p.OH * p.OW;
  // channel tile fastest: the nt blocks that share a token panel run next to each other on ONE XCD (xcd_remap), the panel comes from HBM once
  const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = bid / nt, tile_n = bid - tile_m * nt;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = p.Cin >> 5;

  // ---- X loader: thread -> token row tid >> 2, eight floats at k = 8 (tid & 3) of the chunk ----
  const int xr = tid >> 2, xq = tid & 3;
  const bool xok = m0 + xr < M;
  const float* xsrc = reinterpret_cast<const float*>(p.x) + (long)(xok ? m0 + xr : M - 1) * p.x_ld + xq * 8;
  const int x_dst = xr * 64 + ((xq ^ ((xr >> 1) & 3)) << 4);          // + stage + plane * XPLANE
  // (values, not by-reference lambda captures: the DMA's asm "memory" clobber would pin captured variables to scratch)
  // Two register sets: X(kc+1) waits in one while X(kc+2) is in flight into the other -- an HBM round trip under load (~2.5 us) is longer than
  // a chunk's MFMAs (~0.8 us for the two blocks of a CU), so one chunk of prefetch distance left the loop latency-bound (0.28 of 416.7).
  // The loads are inline asm and the waits hand-counted: the compiler's own wait insertion put s_waitcnt vmcnt(0) in front of every use (loop-carried
  // loads), which drains the set in flight.  Every load is issued unconditionally (row clamped, chunk clamped: a redundant load instead of a
  // branch), so the queue always holds  [set to use | W pieces | other set]  at the top of a chunk and vmcnt(2) is exact; C1_WAIT ties the
  // registers to the wait so that no use can move above it; the loop is drained before the epilogue (a late load must not land in a reused register).
  f32x4 xa0, xa1, xb0, xb1;
#define C1_XLOAD(R0, R1, KC)                                                                                   \
  {                                                                                                             \
    const float* src_ = xsrc + min((KC), nk - 1) * 32;                                                          \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"                \
                 : "=&v"(R0), "=&v"(R1) : "v"(src_) : "memory");                                                 \
  }
#define C1_WAIT(N, R0, R1) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R0), "+v"(R1) :: "memory")
#define C1_XWRITE(STAGE_, R0, R1) c1_xwrite<XPLANE>(smem + (STAGE_) * STAGE + x_dst, R0, R1, xok, p.relu_in)

  // ---- W loader: wave w moves pieces 3 w .. 3 w + 2 of the 24 (plane = piece / 8, rows 16 (piece % 8) ..); the DMA writes lane-linearly, the
  // swizzle is applied on the source side: lane L -> row 16 q + (L >> 2), physical slot L & 3 = logical slot ^ ((row >> 1) & 3)
  const char* wcur[WPW];
  int winc[WPW];
  const char* zero = reinterpret_cast<const char*>(c1_zero_page);
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int pc = wave * WPW + i, pl = pc >> 3, row = (pc & 7) * 16 + (lane >> 2);
    const int j = (lane & 3) ^ ((row >> 1) & 3);
    const int n = n0 + row;
    const char* src = zero;
    if (n < w_rows) src = reinterpret_cast<const char*>(w3 + (size_t)pl * w_bstride + (size_t)n * 32 + j * 8);
    wcur[i] = src;
    winc[i] = src == zero ? 0 : w_rows * 64;                          // chunk-major: the next K chunk is one [w_rows][32] slab further
  }
  const unsigned smem_base = c1_lds_addr(smem);
  auto wissue = [&](int stage) {
    const unsigned dst = smem_base + stage * STAGE + 3 * XPLANE;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int pc = wave * WPW + i;
      c1_glds16(wcur[i], dst + (pc >> 3) * WPLANE + (pc & 7) * 1024);
      wcur[i] += winc[i];
    }
  };
  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int slot = (fg ^ ((fr >> 1) & 3)) << 4;
  const int x_off = (wm * WTM + fr) * 64 + slot;                      // + plane * XPLANE + fm * 1024
  const int w_off = 3 * XPLANE + (wn * WTN + fr) * 64 + slot;         // + plane * WPLANE + fn * 1024

  auto multiply = [&](int s) __attribute__((always_inline)) {
    const char* S = smem + s * STAGE;
    uint4 fw[3][FN], fx[3][FM];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) fw[pl][fn] = *reinterpret_cast<const uint4*>(S + w_off + pl * WPLANE + fn * 1024);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) fx[pl][fm] = *reinterpret_cast<const uint4*>(S + x_off + pl * XPLANE + fm * 1024);
    }
#define C1_TERM(PW, PX)                                                                                                      \
  _Pragma("unroll") for (int fn = 0; fn < FN; ++fn) _Pragma("unroll") for (int fm = 0; fm < FM; ++fm)                       \
      acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[PW][fn]), __builtin_bit_cast(bf16x8, fx[PX][fm]), \
                                                            acc[fn][fm], 0, 0, 0);
    C1_TERM(0, 2) C1_TERM(2, 0) C1_TERM(1, 1) C1_TERM(0, 1) C1_TERM(1, 0) C1_TERM(0, 0)
#undef C1_TERM
  };
  // One chunk: wait for W(kc) (DMA, issued one chunk ago) and X(kc+1) (registers R, issued two chunks ago) -- the loads of X(kc+2), younger than
  // both, stay in flight; the barrier publishes stage kc and retires the reads of stage kc-1; split X(kc+1) into the other stage BEFORE the next DMA
  // is issued (the compiler's own wait for R must not see younger DMA pieces in the queue), issue W(kc+1), reload R with X(kc+3), multiply chunk kc.
#define C1_CHUNK(KC, R0, R1)                                              \
  {                                                                        \
    const int kc_ = (KC), s_ = kc_ & 1;                                    \
    C1_WAIT(2, R0, R1);                                                    \
    c1_barrier();                                                          \
    if (kc_ + 1 < nk) {                                                    \
      C1_XWRITE(s_ ^ 1, R0, R1);                                           \
      wissue(s_ ^ 1);                                                      \
    }                                                                      \
    C1_XLOAD(R0, R1, kc_ + 3)                                              \
    multiply(s_);                                                          \
  }
  // prologue: stage 0 = chunk 0; X(1) in set b, X(2) in set a
  C1_XLOAD(xa0, xa1, 0)
  wissue(0);
  if constexpr (WPW == 6) { C1_WAIT(6, xa0, xa1); } else { C1_WAIT(3, xa0, xa1); }   // X(0) landed (the DMA pieces issued after it may still fly)
  C1_XWRITE(0, xa0, xa1);
  C1_XLOAD(xb0, xb1, 1)
  C1_XLOAD(xa0, xa1, 2)
  for (int kc = 0; kc < nk; kc += 2) {
    C1_CHUNK(kc, xb0, xb1)                                            // even chunk: X(kc+1) waits in set b
    if (kc + 1 < nk) C1_CHUNK(kc + 1, xa0, xa1)
  }
#undef C1_CHUNK
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa0), "+v"(xa1), "+v"(xb0), "+v"(xb1) :: "memory");   // drain the clamped tail loads

  // ---- epilogue: bias -> act -> scale -> residual(s) -> float32 store ----
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) {
    const int n = n0 + wn * WTN + fn * 16 + fg * 4;
    if (n >= p.Cout) continue;
    float4 bias_r = make_float4(0.f, 0.f, 0.f, 0.f), scale_r = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.bias) bias_r = *reinterpret_cast<const float4*>(p.bias + n);
    if (p.scale) scale_r = *reinterpret_cast<const float4*>(p.scale + n);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const int m = m0 + wm * WTM + fm * 16 + fr;
      if (m >= M) continue;
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
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (long)m * p.y_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

}  // namespace

// declared in include/pf_hip.h
extern "C" int pf_conv1x1_split3(const pf_conv_params* p, const void* w3, int w3_rows, void* stream) {
  if (!p || !p->x || !p->y || !w3) return PF_ERR_ARG;
  if (p->dtype != PF_DTYPE_F32 || p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad != 0 || p->shuffle > 1) return PF_ERR_ARG;
  if (p->Cin <= 0 || p->Cin % 32 || p->x_ld % 4 || p->x_ld < p->Cin) return PF_ERR_ARG;
  if (p->Cout <= 0 || p->Cout % 4 || p->y_ld % 4 || w3_rows < p->Cout || w3_rows % 16) return PF_ERR_ARG;
  if ((p->res && p->res_ld % 4) || (p->res2 && p->res2_ld % 4)) return PF_ERR_ARG;
  const long M = (long)p->B * p->OH * p->OW;
  if (M <= 0 || M >= (1L << 31) || (long)w3_rows * 64 >= (1L << 31)) return PF_ERR_ARG;
  if ((reinterpret_cast<size_t>(p->x) | reinterpret_cast<size_t>(p->y) | reinterpret_cast<size_t>(w3)) & 15) return PF_ERR_ARG;
  // PF_C1_BM = 64 | 128 forces a token tile (A/B, tests); default 64 (two blocks per CU)
  static const int force = [] { const char* e = getenv("PF_C1_BM"); return e ? atoi(e) : 0; }();
  const int bm = force == 128 ? 128 : 64;
  const int mt = (int)((M + bm - 1) / bm), nt = (p->Cout + BN - 1) / BN;
  if ((long)mt * nt >= (1L << 31)) return PF_ERR_ARG;
  constexpr int lds64 = 2 * (3 * 64 * 64 + 3 * WPLANE), lds128 = 2 * (3 * 128 * 64 + 3 * WPLANE);
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_split3_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_split3_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds128) == hipSuccess;
  if (!attr_ok) return PF_ERR_LAUNCH;
  const long w_bstride = (long)(p->Cin / 32) * w3_rows * 32;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* w = static_cast<const bf16_t*>(w3);
  const dim3 grid((unsigned)(mt * nt));
  if (bm == 128) hipLaunchKernelGGL(conv1x1_split3_kernel<128>, grid, dim3(512), lds128, st, *p, w, w3_rows, w_bstride, mt, nt);
  else hipLaunchKernelGGL(conv1x1_split3_kernel<64>, grid, dim3(256), lds64, st, *p, w, w3_rows, w_bstride, mt, nt);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}
