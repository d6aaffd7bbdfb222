// Fully fused Winograd F(4x4, 3x3) convolution for the float32 ("exact") mode -- ONE kernel per layer: the transformed input V
// and the transform-domain products M of csrc/winograd.hip's three-step path never exist in HBM.
// Same reference layers: 3x3 / stride 1 / pad 1 convolutions of the guided-fusion U-Net and the DPT head
// (estimator/models/blocks/guided_fusion_model.py:41-48,85-100,129; external/depth_anything/blocks.py:69-92, dpt.py:87-90).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A
//
// Work decomposition.  A block owns a SUPER-TILE of 32 output tiles of 4x4 pixels (4 x 8 or 8 x 4 tiles, picked per layer for the
// least padding) x 64 output channels x ALL 36 transform points; it walks Cin in chunks of 8 channels:
//   raw   : the super-tile's input halo (18 x 34 or 34 x 18 pixels) of SIXTEEN channels = two chunks, brought by LDS-DMA
//           (global_load_lds_dwordx4) into a 2-deep ring; pixel-major, 64 contiguous bytes per pixel, so that a 128-byte line of x is
//           touched twice per 32 channels.  (Round-3 measurement of the first version, which fetched 8 channels = 2 x 16 B per
//           pixel and chunk: the L1 fill of 1792 partial lines per chunk bounded the kernel at 0.52 of the MFMA peak whatever the
//           MFMA work -- 544->32 and 544->544 took the same time per chunk.)  The 16-byte slots of a pixel PAIR are XOR-swizzled
//           with the tile column (source side, like igemm.hip) so that the transform's 4-byte reads are bank-conflict free.
//   V     : B^T d B of one chunk, 36 planes x (32 tiles x 8 ch) float in a bank-rotated order, 2-deep; written by the four
//           "transform" waves from raw
//   MFMA  : wave (pg, half) multiplies planes 9 pg .. 9 pg + 8 for output channels n0 + 32 half .. + 31:
//           D[n][tile] += U_plane[n][c] V_plane[tile][c] on v_mfma_f32_16x16x4_f32 (A = filters, B = tiles), 144 accumulator
//           registers per lane; the filter fragments come straight from global memory (L2) into registers in a pre-packed
//           fragment order (packing.winograd_filters_fused) -- no other wave needs them, so staging them in LDS would buy nothing --
//           as a rolling prefetch: plane p's registers are reloaded for a later plane / the next chunk right after its last MFMA.
//   out   : after the last chunk the 36 planes of a (tile, 4-channel) unit sit in four different waves; they are exchanged through
//           LDS once (one channel half at a time: 36 x 32 x 32 floats = 144 KiB), A^T M A + bias / ReLU / residual(s) run in registers
//           and y is stored once.
// Per 8-channel chunk a block issues 576 MFMAs (4608 cycles per SIMD) against 19 KiB of input halo and 72 KiB of filter fragments.
//
// Measured on the way and NOT kept (round 3, gpurun_out/r3*_wf_time.log, r3j_variants.log): 8 channels (2 x 16 B) per pixel and chunk in the raw
// stage (v1: same speed -- the L1 fill of partial lines was not the limiter); one `s_mov m0` for five DMA pieces through the instruction
// offset, which shifts the global AND the LDS address (+4 %: five back-to-back VMEM issues stall the wave longer than five spread ones);
// the input transform split over all eight waves, output rows 0-2 / 3-5 (+4 %: the DMA waves, already the younger half in the issue
// arbitration, became the critical path); s_setprio around the DMA issue (+-0).
// Wave roles (8 waves, two per SIMD: wave w and w+4).  Waves 0-3 ("transform", channel half 0): planes 0-3, the input transform of
// the next chunk (VALU + LDS only), planes 4-8.  Waves 4-7 ("DMA", channel half 1): 5 LDS-DMA pieces each (half a raw stage per
// chunk) first, then planes 0-8.  The two waves of a SIMD are therefore never both outside their MFMA stream.
// One barrier per chunk.  Global loads and the DMA are issued from inline asm and counted by hand (s_waitcnt vmcnt(N) naming the
// registers it releases, like the hand-counted LDS reads of igemm.hip); tools/asm_vm_audit.py replays the in-order VMEM queue over the
// emitted assembly (tests/test_kernel_resources.py).  tests/wino_fused_model.py replays every index formula below lane by lane.
#include <atomic>
#include "pf_common.h"
#include "../../include/pf_hip.h"

namespace {

constexpr int NT = 32;                        // tiles per super-tile
constexpr int DMA_PIECES = 40;                // 1 KiB each per 16-channel raw stage: 10 per DMA wave, 5 per chunk
constexpr int RAW_STAGE = DMA_PIECES * 1024;  // 40960 >= 18*34 pixels x 64 B
constexpr int V_STAGE = 36 * NT * 8 * 4;      // 36864
constexpr int LDS_V0 = 2 * RAW_STAGE;         // 81920
constexpr int LDS_MAIN = LDS_V0 + 2 * V_STAGE;   // 155648
constexpr int LDS_EPI = 36 * NT * 8 * 16;     // 147456: [36][32 tiles][8 quads] float4 = one channel half of M
constexpr int LDS_TOTAL = LDS_MAIN;
constexpr int U_PLANE = 2 * 1024;             // bytes between planes of the packed filters (2 halves x 64 lanes x 16 B)
constexpr int U_CHUNK = 36 * U_PLANE;         // bytes between 8-channel chunks
static_assert(LDS_EPI <= LDS_MAIN && LDS_TOTAL <= 160 * 1024, "LDS budget");

struct WF {
  const float* x; int x_ld; int B, H, W, Cin;
  const float* up;
  float* y; int y_ld; int Cout;
  const float* bias; int relu; int relu_in;
  const float* res; int res_ld; const float* res2; int res2_ld;
  int TH, TW, nsy, nsx, nsuper, nnb, nkc, gs;
  int dbg;   // only read by the PF_WF_DBG measurement build (make wfdbg; never part of libpf_hip.so; results are WRONG when non-zero):
             // 1 = filter loads always hit chunk 0, 2 = no input transform, 4 = no DMA, 8 / 16 = no MFMA planes in the transform / DMA waves
};
#ifdef PF_WF_DBG
#define WF_DBG(p, bit) ((p).dbg & (bit))
// timeline of ONE block (super-tile * nnb + channel block = 1000 * (dbg >> 8), when dbg & 128): lane 0 of every wave stores s_memtime at the marked points into
// res2 (a long[8][256] buffer in this build): [wave][0] entry, [1] prologue done, [2 + 3c + k] points of chunk c, [250..] epilogue
#define WF_T(idx) do { if (trec && lane == 0) reinterpret_cast<long*>(const_cast<float*>(p.res2))[wave * 256 + (idx)] = (long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define WF_DBG(p, bit) 0
#define WF_T(idx) ((void)0)
#endif

__device__ __attribute__((aligned(256))) unsigned int wf_zero_page[64];

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// LDS-DMA of 16 bytes per lane (see igemm.hip glds16; cdna_hip_programming.md 5.7)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// global -> register load hipcc does not see (no s_waitcnt bookkeeping): the destination is tied ("+v") so that the register
// stays ONE live range through the chunk loop; vm_wait<N> names it again before the first MFMA that reads it.
__device__ __forceinline__ void gload16(f32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(p));
}
template <int N>
__device__ __forceinline__ void vm_wait(f32x4& a) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N>
__device__ __forceinline__ void vm_wait_plain() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() is fence + barrier, and the fence may be lowered
// to vmcnt(0): that would drain the filter prefetch and the DMA that are meant to stay in flight across the barrier.
// The TAG (an assembler comment) keeps the barriers of the two wave roles textually different, so that hipcc does not merge the
// two role paths at a shared barrier: each role's accumulators must be dead once that role has written them to LDS.
template <int TAG = 0>
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier ; role tag %0" ::"n"(TAG) : "memory");
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]   (same formulas as winograd.hip)
__device__ __forceinline__ void bt6(float (&d)[6]) {
  // 12 operations: o1/o2 = (d4 - 4 d2) +- (d3 - 4 d1), o3/o4 = (d4 - d2) +- 2 (d3 - d1)
  const float p = fmaf(-4.f, d[2], d[4]), q = fmaf(-4.f, d[1], d[3]);
  const float r = d[4] - d[2], s = d[3] - d[1];
  const float o0 = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
  const float o5 = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
  d[0] = o0; d[1] = p + q; d[2] = p - q; d[3] = fmaf(2.f, s, r); d[4] = fmaf(-2.f, s, r); d[5] = o5;
}
// A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(const float (&m)[6], float (&o)[4]) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}

// pixel-pair swizzle mask of a raw pixel whose column lies in tile column sx (= rx >> 2): the 8 slots of 16 bytes of a pixel pair
// hold slot (xp << 2 | quad) ^ mask.  For the four tile columns of a 32-lane group and both channel quads of a chunk the transform's
// reads then hit 8 distinct slots = 32 distinct banks.
__device__ __forceinline__ int swz(int sx) { return ((sx & 1) << 2) | (sx & 2); }

// input transform of one (tile, channel) unit: 36 LDS reads at raw + (col[b] ^ jx) + a*ROWB, B^T d B, 36 LDS writes at vout + k*1024
template <int ROWB>
__device__ __forceinline__ void transform_unit(const char* raw, const int (&col)[6], int jx, char* vout, bool relu_in) {
  float d[6][6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const char* cp = raw + (col[b] ^ jx);
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a][b] = *reinterpret_cast<const float*>(cp + a * ROWB);
  }
  if (relu_in) {                                    // (wave-uniform) ResidualConvUnit: ReLU on load
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) d[a][b] = fmaxf(d[a][b], 0.f);
  }
#pragma unroll
  for (int b = 0; b < 6; ++b) {                     // B^T d : down the columns
    float c6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) c6[a] = d[a][b];
    bt6(c6);
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a][b] = c6[a];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {                     // (B^T d) B : along the rows
    bt6(d[i]);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<float*>(vout + (i * 6 + j) * 1024) = d[i][j];
  }
}

template <int SW>     // super-tile width in tiles: 8 -> 4 x 8 tiles (18 x 34 halo), 4 -> 8 x 4 tiles (34 x 18 halo)
__global__ __launch_bounds__(512) void wino_fused_kernel(const WF p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave & 3, half = wave >> 2;
  const unsigned smem_base = lds_addr(smem);

  constexpr int SH = NT / SW, RR = 4 * SH + 2, RCc = 4 * SW + 2, RP = RCc / 2, ROWB = RP * 128, RAW_USED = RR * RCc * 4;
  static_assert(RAW_USED <= DMA_PIECES * 64, "raw stage");
  // ---- block -> (super-tile, channel block): groups of `gs` super-tiles x all channel blocks, super-tiles fastest (consecutive ids
  // share an XCD: its ~32 resident blocks are gs super-tiles x a few channel blocks whose filter streams overlap in L2, and the
  // input halos are re-read by the other channel blocks of the group while they are still L2 / MALL resident) ----
  int sidx, nb;
  {
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int per_group = p.gs * p.nnb;
    const int group = bid / per_group;
    const int first = group * p.gs;
    const int gsz = min(p.nsuper - first, p.gs);
    const int in_g = bid - group * per_group;
    nb = in_g / gsz;
    sidx = first + (in_g - nb * gsz);
  }
  const int b_img = sidx / (p.nsy * p.nsx);
  const int srem = sidx - b_img * (p.nsy * p.nsx);
  const int sty = srem / p.nsx, stx = srem - sty * p.nsx;
  const int n0 = nb * 64;
#ifdef PF_WF_DBG
  const bool trec = (p.dbg & 128) && (int)(sidx * p.nnb + nb) == (p.dbg >> 8) * 1000;
#endif
  WF_T(0);
  const bool active = n0 + half * 32 < p.Cout;          // (wave-uniform) this wave's 32 output channels exist

  // ---- DMA lanes (waves 4-7): pieces hf*20 + 5 pg + i (hf = half of the raw stage issued in this chunk interval); physical slot =
  // piece*64 + lane -> pixel pair, swizzled (pixel parity, channel quad); element offset of the 4 channels or -1 ----
  int doff[10];
  if (half) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int slot = ((i / 5) * 20 + 5 * pg + (i % 5)) * 64 + lane;
      doff[i] = -1;
      if (slot < RAW_USED) {
        const int pairidx = slot >> 3, sp = slot & 7;
        const int ry = pairidx / RP, pr = pairidx - ry * RP;
        const int sg = sp ^ swz(pr >> 1);
        const int rx = 2 * pr + (sg >> 2), q = sg & 3;
        const int iy = 4 * SH * sty - 1 + ry, ix = 4 * SW * stx - 1 + rx;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) doff[i] = ((b_img * p.H + iy) * p.W + ix) * p.x_ld + 4 * q;
      }
    }
  }
  const char* zero = reinterpret_cast<const char*>(wf_zero_page);
  const float* __restrict__ xg = p.x;
  // half `hf` (0 | 1) of the raw stage of channel group G (16 channels = chunks 2G, 2G+1) -> stage G & 1
  auto dma_half = [&](int G, int hf) {
    const unsigned dst = smem_base + (G & 1) * RAW_STAGE + (hf * 20 + 5 * pg) * 1024;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int o = hf ? doff[5 + i] : doff[i];
      const char* src = o >= 0 ? reinterpret_cast<const char*>(xg + (long)o + G * 16) : zero;
      glds16(src, dst + i * 1024);
    }
  };

  // ---- transform lanes (waves 0-3): unit = (tile slot 8 pg + lane/8, channel lane%8 of the chunk); col[b] = byte offset of raw
  // pixel (4 sy, 4 sx + b), channel quad h of an EVEN chunk (odd chunks: ^ 32), + 4 cc ----
  const int tc = lane & 7, tsl = 8 * pg + (lane >> 3);
  int tcol[6];
  {
    const int sy = tsl / SW, sx = tsl % SW, h = tc >> 2, cc = tc & 3;
    const int m0 = swz(sx), m1 = swz(sx + 1);
#pragma unroll
    for (int b = 0; b < 6; ++b)
      tcol[b] = ((((4 * sy * RP + 2 * sx + (b >> 1)) * 8) + ((((b & 1) << 2) | h) ^ (b < 4 ? m0 : m1))) * 16) + cc * 4;
  }
  // V image of a plane (1 KiB): dword (c >> 1) * 64 + (tile >> 4) * 32 + (((tile & 15) + 4 (c >> 1)) & 15) * 2 + (c & 1).  The rotation by
  // 4 (c >> 1) makes BOTH sides conflict free: the transform's 4-byte stores (32 lanes = 8 channels x 4 tiles -> 32 banks) and the
  // MFMA waves' 8-byte fragment reads, which hipcc merges into ds_read2_b64 = 16-lane groups over 32 banks (16 tiles x 2 dwords).
  // (Round-3 PMC of the first layout, 32 bytes per tile: 58 % of the LDS cycles were bank conflicts -- 4-way on every fragment read.)
  const int t_wr = ((tc >> 1) * 64 + (tsl >> 4) * 32 + ((((tsl & 15) + 4 * (tc >> 1)) & 15) * 2) + (tc & 1)) * 4;
  const bool lo = p.relu_in != 0;

  // ---- MFMA lanes: B fragment (tile r, channels 2 g4, 2 g4 + 1) of plane 9 pg + P, tile group tg ----
  const int r = lane & 15, g4 = lane >> 4;
  const int b_rd = (g4 * 64 + ((r + 4 * g4) & 15) * 2) * 4 + 9 * pg * 1024;      // tile group tg: + 128 bytes
  const char* ubase = reinterpret_cast<const char*>(p.up) + ((((long)nb * p.nkc) * 36 + 9 * pg) * 2 + half) * 1024 + lane * 16;

  f32x4 acc[9][2][2];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[i][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 u[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one plane P with its filter fragment in u[S] and its B fragments already in BC0 / BC1: request the B fragments of plane P+1 (so their
  // LDS latency hides behind this plane's MFMAs), wait for the filter fragment, 8 MFMAs (4 accumulators x 2 k-steps; consecutive MFMAs
  // hit different accumulators: 40-cycle dependent latency against the 32-cycle issue), then reload u[S] from NEXT
#define WF_BLOAD(P, B0, B1)                                                                              \
  B0 = *reinterpret_cast<const float2*>(vcur + (P) * 1024);                                              \
  B1 = *reinterpret_cast<const float2*>(vcur + (P) * 1024 + 128);
#define WF_PLANE_(P, S, WAITN, NEXT, BC0, BC1, PREFETCH)                                                 \
  {                                                                                                      \
    PREFETCH                                                                                             \
    vm_wait<WAITN>(u[S]);                                                                                \
    acc[P][0][0] = mfma4(u[S][0], BC0.x, acc[P][0][0]);                                                  \
    acc[P][1][0] = mfma4(u[S][0], BC1.x, acc[P][1][0]);                                                  \
    acc[P][0][1] = mfma4(u[S][2], BC0.x, acc[P][0][1]);                                                  \
    acc[P][1][1] = mfma4(u[S][2], BC1.x, acc[P][1][1]);                                                  \
    acc[P][0][0] = mfma4(u[S][1], BC0.y, acc[P][0][0]);                                                  \
    acc[P][1][0] = mfma4(u[S][1], BC1.y, acc[P][1][0]);                                                  \
    acc[P][0][1] = mfma4(u[S][3], BC0.y, acc[P][0][1]);                                                  \
    acc[P][1][1] = mfma4(u[S][3], BC1.y, acc[P][1][1]);                                                  \
    NEXT;                                                                                                \
  }
  // even planes hold their B fragments in (be0, be1), odd planes in (bo0, bo1)
#define WF_PLANE(P, S, WAITN, NEXT) WF_PLANE_E##P(P, S, WAITN, NEXT)
#define WF_PLANE_E0(P, S, W, N) WF_PLANE_(P, S, W, N, be0, be1, WF_BLOAD(1, bo0, bo1))
#define WF_PLANE_E1(P, S, W, N) WF_PLANE_(P, S, W, N, bo0, bo1, WF_BLOAD(2, be0, be1))
#define WF_PLANE_E2(P, S, W, N) WF_PLANE_(P, S, W, N, be0, be1, WF_BLOAD(3, bo0, bo1))
#define WF_PLANE_E3(P, S, W, N) WF_PLANE_(P, S, W, N, bo0, bo1, WF_BLOAD(4, be0, be1))
#define WF_PLANE_E4(P, S, W, N) WF_PLANE_(P, S, W, N, be0, be1, WF_BLOAD(5, bo0, bo1))
#define WF_PLANE_E5(P, S, W, N) WF_PLANE_(P, S, W, N, bo0, bo1, WF_BLOAD(6, be0, be1))
#define WF_PLANE_E6(P, S, W, N) WF_PLANE_(P, S, W, N, be0, be1, WF_BLOAD(7, bo0, bo1))
#define WF_PLANE_E7(P, S, W, N) WF_PLANE_(P, S, W, N, bo0, bo1, WF_BLOAD(8, be0, be1))
#define WF_PLANE_E8(P, S, W, N) WF_PLANE_(P, S, W, N, be0, be1, )
  float2 be0, be1, bo0, bo1;
#define WF_NONE ((void)0)

  // ---- epilogue: exchange one channel half at a time through LDS [36][32 tiles][8 quads ^ (tile & 7)] float4 ----
  auto epi_write = [&]() {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
          *reinterpret_cast<f32x4*>(smem + ((((9 * pg + i) * 32 + tg * 16 + r) * 8) + ((cg * 4 + g4) ^ (r & 7))) * 16) = acc[i][tg][cg];
  };
  // read side of the exchange.  Unit of lane l in wave pg: tile 8 pg + l / 8, channel quad l % 8 of this half -> the eight lanes of a
  // tile store 128 contiguous bytes per pixel (whole lines of y; the first version stored 64-byte halves of 16 different pixels per
  // instruction and its store tail took 10.8k cycles per half).  First stage: A^T m down the columns, all 36 reads.
  float tq[4][6][4];
  const int etile = 8 * pg + (lane >> 3), equad = lane & 7;
  auto epi_read = [&]() {
    const char* mp = smem + (etile * 8 + (equad ^ (etile & 7))) * 16;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x4 m[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(mp + (i * 6 + j) * (32 * 8 * 16));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float col[6] = {m[0][e], m[1][e], m[2][e], m[3][e], m[4][e], m[5][e]};
        float o[4];
        at6(col, o);
#pragma unroll
        for (int po = 0; po < 4; ++po) tq[po][j][e] = o[po];
      }
      __builtin_amdgcn_sched_barrier(0);      // keep hipcc from hoisting all 36 reads (144 registers) above the first column
    }
  };
  // second stage: along the rows, bias / ReLU / residual(s), stores
  auto epi_store = [&]() {
    const int n = n0 + half * 32 + 4 * equad;
    const int ty = SH * sty + etile / SW, tx = SW * stx + etile % SW, b = b_img;
    if (ty >= p.TH || tx >= p.TW || n >= p.Cout) return;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
    const float be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int po = 0; po < 4; ++po) {
      const int oy = 4 * ty + po;
      float o[4][4];                                   // [channel e][column qo]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float row[6] = {tq[po][0][e], tq[po][1][e], tq[po][2][e], tq[po][3][e], tq[po][4][e], tq[po][5][e]};
        at6(row, o[e]);
      }
      if (oy >= p.H) continue;
#pragma unroll
      for (int qo = 0; qo < 4; ++qo) {
        const int ox = 4 * tx + qo;
        if (ox >= p.W) continue;
        const long pix = ((long)b * p.H + oy) * p.W + ox;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = o[e][qo] + be[e];
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.res) {
          const float4 a = *reinterpret_cast<const float4*>(p.res + pix * p.res_ld + n);
          v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        }
        if (p.res2 && !WF_DBG(p, 128)) {
          const float4 a = *reinterpret_cast<const float4*>(p.res2 + pix * p.res2_ld + n);
          v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        }
        *reinterpret_cast<float4*>(p.y + pix * p.y_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  const int nkc = p.nkc;
  if (!half) {
    // ================= transform waves (0-3): all nine filter fragments of a chunk stay in registers; plane P's is re-requested for the next
    // chunk right after its last MFMA: VMEM queue when plane P waits = [u_c(P), u_c(P+1..8), u_{c+1}(0..P-1)] -> 8 younger loads.
    // (A 3-deep ring u[P % 3] with vmcnt(2) measured 0.5 % slower at gs 8 and 3-5 % slower at larger block groups: the two planes
    // between request and use are shorter than the L2 latency whenever this wave has the matrix pipe to itself.) =================
#pragma unroll
    for (int i = 0; i < 9; ++i) gload16(u[i], ubase + i * U_PLANE);
#pragma unroll
    for (int i = 0; i < 9; ++i) vm_wait<0>(u[i]);
    lds_barrier<1>();
    transform_unit<ROWB>(smem, tcol, 0, smem + LDS_V0 + t_wr, lo);
    lds_barrier<1>();
    WF_T(1);
    for (int c = 0; c < nkc - 1; ++c) {
      const char* vcur = smem + LDS_V0 + (c & 1) * V_STAGE + b_rd;
      const char* un = ubase + (long)(c + 1) * U_CHUNK;
      if (c < 60) WF_T(2 + 4 * c);
      WF_BLOAD(0, be0, be1)
      WF_PLANE(0, 0, 8, gload16(u[0], un))
      WF_PLANE(1, 1, 8, gload16(u[1], un + 1 * U_PLANE))
      WF_PLANE(2, 2, 8, gload16(u[2], un + 2 * U_PLANE))
      WF_PLANE(3, 3, 8, gload16(u[3], un + 3 * U_PLANE))
      if (c < 60) WF_T(3 + 4 * c);
      transform_unit<ROWB>(smem + (((c + 1) >> 1) & 1) * RAW_STAGE, tcol, ((c + 1) & 1) << 5, smem + LDS_V0 + ((c + 1) & 1) * V_STAGE + t_wr, lo);
      if (c < 60) WF_T(4 + 4 * c);
      WF_PLANE(4, 4, 8, gload16(u[4], un + 4 * U_PLANE))
      WF_PLANE(5, 5, 8, gload16(u[5], un + 5 * U_PLANE))
      WF_PLANE(6, 6, 8, gload16(u[6], un + 6 * U_PLANE))
      WF_PLANE(7, 7, 8, gload16(u[7], un + 7 * U_PLANE))
      WF_PLANE(8, 8, 8, gload16(u[8], un + 8 * U_PLANE))
      if (c < 60) WF_T(5 + 4 * c);
      lds_barrier<1>();
    }
    {
      const char* vcur = smem + LDS_V0 + ((nkc - 1) & 1) * V_STAGE + b_rd;
#pragma unroll
      for (int i = 0; i < 9; ++i) vm_wait<0>(u[i]);
      WF_BLOAD(0, be0, be1)
      WF_PLANE(0, 0, 0, WF_NONE) WF_PLANE(1, 1, 0, WF_NONE) WF_PLANE(2, 2, 0, WF_NONE) WF_PLANE(3, 3, 0, WF_NONE)
      WF_PLANE(4, 4, 0, WF_NONE) WF_PLANE(5, 5, 0, WF_NONE) WF_PLANE(6, 6, 0, WF_NONE) WF_PLANE(7, 7, 0, WF_NONE)
      WF_PLANE(8, 8, 0, WF_NONE)
    }
    WF_T(249);
    lds_barrier<1>();                                      // every wave has finished reading V; no DMA is in flight
    WF_T(250);
    epi_write();                                           // channel half 0 first
    lds_barrier<1>();
    WF_T(251);
    epi_read();
    lds_barrier<1>();                                      // the exchange buffer is free: the DMA waves write channel half 1 ...
    lds_barrier<1>();                                      // (their write -> read barrier)
    WF_T(252);
    epi_store();                                           // ... and both halves' store tails run side by side
    WF_T(253);
  } else {
    // ================= DMA waves: all nine fragments of a chunk stay in registers; plane P's is re-requested for the next chunk right
    // after its last MFMA.  VMEM queue when plane P waits: [u_c(P), u_c(P+1..8), DMA(c) x 5, u_{c+1}(0..P-1)] -> 13 younger
    // (DMA(c-1) was issued before u_c(P)) =================
    const int ng = nkc >> 1;
    dma_half(0, 0);
    dma_half(0, 1);
    dma_half(min(1, ng - 1), 0);          // first half of group 1 (the 'interval -1' share)
    if (active) {
#pragma unroll
      for (int i = 0; i < 9; ++i) gload16(u[i], ubase + i * U_PLANE);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) vm_wait<0>(u[i]);
    lds_barrier<2>();
    lds_barrier<2>();
    WF_T(1);
    if (active) {
      for (int c = 0; c < nkc - 1; ++c) {
        const char* vcur = smem + LDS_V0 + (c & 1) * V_STAGE + b_rd;
        const char* un = ubase + (long)(WF_DBG(p, 1) ? 0 : c + 1) * U_CHUNK;
        if (c < 60) WF_T(2 + 4 * c);
        // the DMA waves are the younger half of the block: without priority their ~50 issue-side instructions wait behind every MFMA of
        // the co-resident transform wave (s_memtime timeline, round 3: 2250 cycles for five pieces, and 750 cycles per chunk in which
        // NEITHER wave of the SIMD issued an MFMA)
        __builtin_amdgcn_s_setprio(3);
        if (!WF_DBG(p, 4)) dma_half(min((c + 3) >> 1, ng - 1), (c + 1) & 1);   // group (c+3)/2: first half in odd c, second half in even c (a harmless re-load at the end)
        __builtin_amdgcn_s_setprio(0);
        if (c < 60) WF_T(3 + 4 * c);
        if (!WF_DBG(p, 16)) {
        WF_BLOAD(0, be0, be1)
        WF_PLANE(0, 0, 13, gload16(u[0], un))
        WF_PLANE(1, 1, 13, gload16(u[1], un + 1 * U_PLANE))
        WF_PLANE(2, 2, 13, gload16(u[2], un + 2 * U_PLANE))
        WF_PLANE(3, 3, 13, gload16(u[3], un + 3 * U_PLANE))
        WF_PLANE(4, 4, 13, gload16(u[4], un + 4 * U_PLANE))
        WF_PLANE(5, 5, 13, gload16(u[5], un + 5 * U_PLANE))
        WF_PLANE(6, 6, 13, gload16(u[6], un + 6 * U_PLANE))
        WF_PLANE(7, 7, 13, gload16(u[7], un + 7 * U_PLANE))
        WF_PLANE(8, 8, 13, gload16(u[8], un + 8 * U_PLANE))
        }
        if (WF_DBG(p, 16)) vm_wait_plain<0>();
        if (c < 60) WF_T(4 + 4 * c);
        vm_wait_plain<9>();                // DMA(c) has landed: only the nine reloads are younger
        if (c < 60) WF_T(5 + 4 * c);
        lds_barrier<2>();
      }
    } else {
      // the upper channel half does not exist (last channel block of a layer with Cout % 64 == 32): this wave only feeds the DMA
      for (int c = 0; c < nkc - 1; ++c) {
        dma_half(min((c + 3) >> 1, ng - 1), (c + 1) & 1);
        vm_wait_plain<0>();
        lds_barrier<2>();
      }
    }
    {
      const char* vcur = smem + LDS_V0 + ((nkc - 1) & 1) * V_STAGE + b_rd;
#pragma unroll
      for (int i = 0; i < 9; ++i) vm_wait<0>(u[i]);     // every fragment of the last chunk and the last DMA
      if (active) {
        WF_BLOAD(0, be0, be1)
        WF_PLANE(0, 0, 0, WF_NONE) WF_PLANE(1, 1, 0, WF_NONE) WF_PLANE(2, 2, 0, WF_NONE) WF_PLANE(3, 3, 0, WF_NONE)
        WF_PLANE(4, 4, 0, WF_NONE) WF_PLANE(5, 5, 0, WF_NONE) WF_PLANE(6, 6, 0, WF_NONE) WF_PLANE(7, 7, 0, WF_NONE)
        WF_PLANE(8, 8, 0, WF_NONE)
      }
    }
    WF_T(249);
    lds_barrier<2>();
    lds_barrier<2>();
    lds_barrier<2>();
    WF_T(250);
    if (active) epi_write();                               // channel half 1
    lds_barrier<2>();
    WF_T(251);
    if (active) {
      epi_read();
      epi_store();
    }
    WF_T(252);
  }
#undef WF_PLANE
#undef WF_PLANE_
#undef WF_BLOAD
#undef WF_NONE
}

template <int SW>
int launch_sw(const WF& a, hipStream_t st) {
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<SW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(wino_fused_kernel<SW>, dim3((unsigned)((long)a.nsuper * a.nnb)), dim3(512), LDS_TOTAL, st, a);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

// `shape`: 0 = pick the super-tile shape with fewer blocks (less padding), 8 = 4 x 8 tiles, 4 = 8 x 4 tiles
int launch(const pf_conv_params* p, const void* up, int nnb, int gs, int shape, int dbg, hipStream_t st) {
  WF a;
  a.dbg = dbg;
  a.x = static_cast<const float*>(p->x); a.x_ld = p->x_ld; a.B = p->B; a.H = p->H; a.W = p->W; a.Cin = p->Cin;
  a.up = static_cast<const float*>(up);
  a.y = static_cast<float*>(p->y); a.y_ld = p->y_ld; a.Cout = p->Cout;
  a.bias = p->bias; a.relu = p->act == PF_ACT_RELU ? 1 : 0; a.relu_in = p->relu_in;
  a.res = static_cast<const float*>(p->res); a.res_ld = p->res_ld;
  a.res2 = static_cast<const float*>(p->res2); a.res2_ld = p->res2_ld;
  a.TH = (p->H + 3) / 4; a.TW = (p->W + 3) / 4;
  const long n8 = (long)((a.TH + 3) / 4) * ((a.TW + 7) / 8), n4 = (long)((a.TH + 7) / 8) * ((a.TW + 3) / 4);
  const int sw = shape == 8 || shape == 4 ? shape : (n4 < n8 ? 4 : 8);
  const int sh = NT / sw;
  a.nsy = (a.TH + sh - 1) / sh; a.nsx = (a.TW + sw - 1) / sw;
  a.nsuper = p->B * a.nsy * a.nsx;
  a.nnb = nnb;
  a.nkc = p->Cin / 8;
  a.gs = gs < 1 ? 1 : gs;
  return sw == 8 ? launch_sw<8>(a, st) : launch_sw<4>(a, st);
}

}  // namespace

extern "C" int pf_conv_winograd_fused_supported(const pf_conv_params* p) {
  if (!p || p->dtype != PF_DTYPE_F32 || p->KH != 3 || p->KW != 3 || p->stride != 1 || p->pad != 1 || p->shuffle > 1 || p->scale) return 0;
  if (p->OH != p->H || p->OW != p->W || p->Cin % 16 || p->Cin < 32 || p->Cout % 4 || p->Cout <= 0) return 0;
  if (p->act != PF_ACT_NONE && p->act != PF_ACT_RELU) return 0;
  if (p->x_ld % 4 || p->y_ld % 4 || (p->res && p->res_ld % 4) || (p->res2 && p->res2_ld % 4)) return 0;
  const long pix = (long)p->B * p->H * p->W;
  if (pix * p->x_ld >= (1L << 31) || (long)p->B * ((p->H + 15) / 16) * ((p->W + 15) / 16) * ((p->Cout + 63) / 64) >= (1L << 31)) return 0;
  return 1;
}

extern "C" int pf_conv_winograd_fused(const pf_conv_params* p, const void* up, int nnb, int gs, void* stream) {
  if (!p || !up || !p->x || !p->y) return PF_ERR_ARG;
  if (!pf_conv_winograd_fused_supported(p) || nnb != (p->Cout + 63) / 64) return PF_ERR_ARG;
  // gs: low 16 bits = super-tiles per block group; bits 16-19 = forced super-tile shape (8 | 4, 0 = automatic), bits 20.. = the
  // timing-decomposition switches of WF::dbg -- tuning aids
  return launch(p, up, nnb, gs & 0xffff, (gs >> 16) & 0xf, gs >> 20, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pf_conv_winograd_fused_timed(const pf_conv_params* p, const void* up, int nnb, int gs, int iters, float* ms, void* stream) {
  if (!ms || iters <= 0) return PF_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = pf_conv_winograd_fused(p, up, nnb, gs, stream);   // warm-up
  if (rc != PF_OK) return rc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < iters && rc == PF_OK; ++i) rc = pf_conv_winograd_fused(p, up, nnb, gs, stream);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  *ms = t / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return rc;
}
