// 1x1 convolution / linear layer with FLOAT32 activations on the bf16 matrix cores (round 6): the split-precision product of
// csrc/gemm_split3.hip -- x = x_h + x_m + x_l, w = w_h + w_m + w_l in bf16, the six leading partial products accumulated in float32, smallest
// first -- for layers whose producers write plain float32 NHWC tensors (the DPT / metric-bins heads' 1x1 convolutions, localbins_layers.py:99-117,
// attractor.py:156-161, the G2L Swin linears, swin_layers.py:120-128,133-164).  Until round 5 these ran on v_mfma_f32_16x16x4_f32 (1/16 of the bf16
// rate, 0.44-0.70 of THAT pipe); a stand-alone split pass in front of pf_gemm_split3 would move as many bytes as the f32 kernel loses, so the split
// happens HERE: the float32 token rows go to LDS as they are (LDS-DMA), and a wave splits its B fragments in registers right before its MFMAs
// (33 VALU instructions per 8 floats); the weights arrive pre-split (packing.pack_conv_split3, chunk-major) by LDS-DMA as well.
//
//   D[n][m] = sum_k W[n][k] X[m][k]     MFMA A = weight rows, B = token rows; a lane's four accumulator registers = four consecutive channels
//                                        of one token (float4 stores along the channel axis), exactly as in gemm_split3.hip / igemm.hip
// Tile 64 (or 128) tokens x 128 channels, K chunks of 32, four (eight) waves of 32 x 64 (FM = 2, FN = 4: 48 MFMAs per chunk and wave).
// LDS: a ring of THREE float32 token stages (BM rows x 128 B; the float4 k = 4 u .. 4 u + 3 of row r in stored slot sg = (u >> 1) + 4 (u & 1) at
// sg ^ ((r >> 1) & 7)) and TWO weight stages (3 planes x 128 rows x 64 B; slot g of row r at g ^ ((r >> 1) & 3)): every ds_read_b128 of a fragment touches
// 16 distinct 16-byte positions of the 256-byte bank window in each of the hardware's 16-lane groups -- which pair even-g lanes of rows {0-3, 12-15} with
// odd-g lanes of rows {4-11}: the natural slot order u = 2 g + h collides there (tests/test_conv1x1_lds_layout_cpu.py replays both).  The DMA writes
// lane-linearly, so both swizzles are applied on the source side.
// EVERY load is an LDS-DMA piece, so the hand-counted s_waitcnt vmcnt(N) sees ONE in-order queue.  (The first form of this kernel loaded the tokens
// into registers -- split by the loading thread, planes written to LDS -- beside the weights' DMA pieces, with vmcnt(2) leaving the younger register
// set in flight: fast, and WRONG on cold caches -- DMA pieces and register loads do not retire in issue order with respect to each other, the
// count said "two left" while a needed piece was still flying.  tools/c1_debug.py + the PF_C1_DBG build, profiles/r6_conv1x1_split3.md.)
// PERSISTENT tile walk: two blocks per CU (72 KiB each at BM = 64); logical block L (xcd_remap: consecutive L on one XCD) owns channel tile L % nt and
// walks the token tiles L / nt, + nslots, ...; the K chunks of consecutive tiles form ONE stream through the rings, so the first chunks of tile t+1 are
// in flight under the last MFMAs and the epilogue of tile t.  nslots == mt is the one-tile-per-block form (PF_C1_PERSIST=0).
// Pipeline, ONE barrier per chunk g: wait for W(g) (issued one chunk ago) and X(g) (two chunks ago) -- the pieces of X(g+1), younger than both, stay in
// flight; the barrier publishes them and retires the reads of chunk g-1; issue W(g+1) and X(g+2) into the slots chunk g-1 left; read the fragments,
// split the token fragments, multiply; after a tile's last chunk its epilogue (as pf_conv: (act(v + bias) * scale) + res + res2, float32 out).
// RULE for a hand-counted wait: what may stay in flight must be of the same kind as, and younger than, everything the wait is for.
// (Measured and NOT kept, profiles/r6_conv1x1_split3.md: the 128 x 128-tile persistent PING-PONG kernel of gemm_split3.hip with a float32-token stage and the
// split in a wave's load phase -- correct, and 0.8x of this kernel on every layer: DMA issue + split + fragment reads do not fit beside the partner
// group's 48 MFMAs at K = 128 .. 1024.)
#include <atomic>
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
constexpr int WSTAGE = 3 * BN * 64;               // bytes of one weight stage: planes h | m | l of 128 rows x 64 B

template <int BM>
__global__ __launch_bounds__(4 * BM, 2) void conv1x1_split3_kernel(const pf_conv_params p, const bf16_t* __restrict__ w3, int w_rows, long w_bstride,
                                                                   int mt, int nt, int nslots) {
  constexpr int WM = BM / WTM, NW = WM * WN;
  constexpr int XSTAGE = BM * 128;                 // bytes of one float32 token stage
  constexpr int WBASE = 3 * XSTAGE;                // the two weight stages follow the three token stages
  constexpr int WPW = 24 / NW;                     // weight DMA pieces (16 rows x 64 B) per wave and chunk: 3 planes x 8 pieces
  constexpr int XPW = (BM / 8) / NW;               // token DMA pieces (8 rows x 128 B) per wave and chunk
  static_assert(24 % NW == 0 && (BM / 8) % NW == 0 && XPW == 2, "pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = p.B * p.OH * p.OW;
  const int L = xcd_remap((int)blockIdx.x, (int)gridDim.x);           // gridDim.x == nslots * nt
  const int mslot = L / nt, tile_n = L - mslot * nt;
  const int n0 = tile_n * BN;
  const int nk = p.Cin >> 5;
  const int ntl = (mt - mslot + nslots - 1) / nslots;                 // token tiles of this block (>= 1: the host keeps nslots <= mt)
  const int total = ntl * nk;                                         // K chunks of the whole walk
#ifdef PF_C1_DBG           // timing decomposition (results wrong by construction), tools/conv1x1_time.py decomp: bit 0 no MFMA, 1 no token DMA after the
  const int dbg = p.pad;   // prologue, 2 no weight DMA after the prologue, 3 no split (raw bits as planes), 4 one stage re-read, 5 no barrier, 6 every wait vmcnt(0)
#define C1_DBG(bit) (dbg & (bit))
#else
#define C1_DBG(bit) 0
#endif
  const char* zero = reinterpret_cast<const char*>(c1_zero_page);
  const unsigned smem_base = c1_lds_addr(smem);

  // ---- token loader: wave w moves pieces XPW w, XPW w + 1 of the BM / 8 (8 rows x 128 B each); lane L -> row 8 piece + (L >> 3), physical slot
  // L & 7 = stored slot ^ ((row >> 1) & 7).  Rows beyond M read the zero page.  (lt, lk) = tile of the walk and chunk of the NEXT token issue.
  int lt = 0, lk = 0;
  const char* xcur[XPW];
  int xoff[XPW];                                                      // byte offset of the lane's source inside a token row's chunk
  int xrow[XPW];
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    const int pc = wave * XPW + i, row = pc * 8 + (lane >> 3);
    xrow[i] = row;
    const int sg = (lane & 7) ^ ((row >> 1) & 7);                      // stored slot -> the float4 k = 4 u .. 4 u + 3 it holds: u = 2 (sg & 3) + (sg >> 2)
    xoff[i] = ((((sg & 3) << 1) | (sg >> 2)) << 4);
  }
  auto xtile = [&](int t) __attribute__((always_inline)) {            // the lane's source rows of tile t of the walk
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      const long m = (long)(mslot + t * nslots) * BM + xrow[i];
      xcur[i] = m < M ? reinterpret_cast<const char*>(p.x) + (m * p.x_ld) * 4 + xoff[i] : zero;
    }
  };
  xtile(0);
  int issued_x = 0, issued_w = 0;
  auto xissue = [&](int stage) __attribute__((always_inline)) {       // the next token chunk of the walk -> token stage `stage`
    if (!(C1_DBG(2) && issued_x > 2)) {
      const unsigned dst = smem_base + stage * XSTAGE;
#pragma unroll
      for (int i = 0; i < XPW; ++i) c1_glds16(xcur[i] == zero ? zero : xcur[i] + lk * 128, dst + (wave * XPW + i) * 1024);
    }
    ++issued_x;
    if (lk + 1 < nk) ++lk;
    else if (lt + 1 < ntl) {
      lk = 0;
      ++lt;
      xtile(lt);
    }
  };

  // ---- weight loader: wave w moves pieces WPW w .. of the 24 (plane = piece / 8, rows 16 (piece % 8) ..); lane L -> row 16 q + (L >> 2), physical
  // slot L & 3 = logical slot ^ ((row >> 1) & 3).  The channel tile is the same for every tile of the walk: chunk wk of the weight panel, round and round.
  const char* wbase[WPW];
  int winc[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int pc = wave * WPW + i, pl = pc >> 3, row = (pc & 7) * 16 + (lane >> 2);
    const int j = (lane & 3) ^ ((row >> 1) & 3);
    const int n = n0 + row;
    const char* src = zero;
    if (n < w_rows) src = reinterpret_cast<const char*>(w3 + (size_t)pl * w_bstride + (size_t)n * 32 + j * 8);
    wbase[i] = src;
    winc[i] = src == zero ? 0 : w_rows * 64;                          // chunk-major: the next K chunk is one [w_rows][32] slab further
  }
  int wk = 0;
  auto wissue = [&](int stage) __attribute__((always_inline)) {
    if (!(C1_DBG(4) && issued_w)) {
      const unsigned dst = smem_base + WBASE + stage * WSTAGE;
#pragma unroll
      for (int i = 0; i < WPW; ++i) {
        const int pc = wave * WPW + i;
        c1_glds16(wbase[i] + (long)wk * winc[i], dst + (pc >> 3) * (BN * 64) + (pc & 7) * 1024);
      }
    }
    issued_w = 1;
    wk = wk + 1 < nk ? wk + 1 : 0;
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // token fragment: row wm * 32 + fm * 16 + fr, k = 8 fg .. 8 fg + 3 in stored slot fg, k = 8 fg + 4 .. 8 fg + 7 in stored slot fg + 4; the swizzle term
  // is per lane
  const int x_off = (wm * WTM + fr) * 128 + ((fg ^ ((fr >> 1) & 7)) << 4);              // + fm * 2048; second half at ^ 64
  const int w_off = (wn * WTN + fr) * 64 + ((fg ^ ((fr >> 1) & 3)) << 4);               // + plane * BN * 64 + fn * 1024
  // the block's channel tile never changes: bias in registers for the whole walk
  float4 bias_r[FN];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) {
    const int n = n0 + wn * WTN + fn * 16 + fg * 4;
    bias_r[fn] = (p.bias && n < p.Cout) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  c1_vm_wait<0>();                                                    // (the bias loads leave the counter before the hand-counted DMA queue starts)

  auto multiply = [&](int xs, int ws) __attribute__((always_inline)) {
    const char* XS = smem + xs * XSTAGE;
    const char* WS = smem + WBASE + ws * WSTAGE;
    if (C1_DBG(16)) { XS = smem; WS = smem + WBASE; }
    uint4 fw[3][FN], fx[3][FM];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) fw[pl][fn] = *reinterpret_cast<const uint4*>(WS + w_off + pl * (BN * 64) + fn * 1024);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const float4 a = *reinterpret_cast<const float4*>(XS + x_off + fm * 2048);
      const float4 b = *reinterpret_cast<const float4*>(XS + (x_off ^ 64) + fm * 2048);
      float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      if (C1_DBG(8)) {
        fx[0][fm] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        fx[1][fm] = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
        fx[2][fm] = fx[0][fm];
        continue;
      }
      if (p.relu_in) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      uint32_t h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
      fx[0][fm] = make_uint4(h[0], h[1], h[2], h[3]);
      fx[1][fm] = make_uint4(m[0], m[1], m[2], m[3]);
      fx[2][fm] = make_uint4(l[0], l[1], l[2], l[3]);
    }
#define C1_TERM(PW, PX)                                                                                                      \
  _Pragma("unroll") for (int fn = 0; fn < FN; ++fn) _Pragma("unroll") for (int fm = 0; fm < FM; ++fm)                       \
      acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[PW][fn]), __builtin_bit_cast(bf16x8, fx[PX][fm]), \
                                                            acc[fn][fm], 0, 0, 0);
    if (!C1_DBG(1)) { C1_TERM(0, 2) C1_TERM(2, 0) C1_TERM(1, 1) C1_TERM(0, 1) C1_TERM(1, 0) C1_TERM(0, 0) }
    else { _Pragma("unroll") for (int fn = 0; fn < FN; ++fn) _Pragma("unroll") for (int fm = 0; fm < FM; ++fm) _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) { acc[fn][fm][0] += __uint_as_float(fw[pl][fn].x ^ fx[pl][fm].x); } }
#undef C1_TERM
  };
  // ---- epilogue of one tile: bias -> act -> scale -> residual(s) -> float32 store; the accumulators start the next tile at zero ----
  auto epilogue = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int n = n0 + wn * WTN + fn * 16 + fg * 4;
      const bool nok = n < p.Cout;
      float4 scale_r = make_float4(1.f, 1.f, 1.f, 1.f);
      if (p.scale && nok) scale_r = *reinterpret_cast<const float4*>(p.scale + n);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int m = m0 + wm * WTM + fm * 16 + fr;
        float v[4] = {acc[fn][fm][0] + bias_r[fn].x, acc[fn][fm][1] + bias_r[fn].y, acc[fn][fm][2] + bias_r[fn].z, acc[fn][fm][3] + bias_r[fn].w};
        acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!nok || m >= M) continue;
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
  };

  // prologue: X(0), W(0), X(1) in flight
  xissue(0);
  wissue(0);
  if (total > 1) xissue(1);
  int ck = 0, ct = 0;
  for (int g = 0, xs = 0; g < total; ++g) {
    // queue, oldest first:  ... X(g) | W(g) | X(g+1) [| the stores of an epilogue]: the XPW pieces of X(g+1) may stay in flight (none were issued when
    // g + 1 == total).  Safe with stores behind them too: the pieces retire in issue order among themselves, so while a needed piece is out so are both
    // pieces of X(g+1) and the count is above XPW whatever the stores do (they retire out of order with loads; late ones only make the wait longer).
    if (g + 1 >= total || C1_DBG(64)) c1_vm_wait<0>();
    else c1_vm_wait<XPW>();
    if (!C1_DBG(32)) c1_barrier();
    if (g + 1 < total) wissue((g + 1) & 1);
    if (g + 2 < total) xissue(xs == 0 ? 2 : xs - 1);                  // stage (g + 2) % 3
    multiply(xs, g & 1);
    xs = xs == 2 ? 0 : xs + 1;
    if (ck + 1 < nk) ++ck;
    else {
      epilogue((mslot + ct * nslots) * BM);
      ck = 0;
      ++ct;
    }
  }
}

}  // namespace

// declared in include/pf_hip.h
extern "C" int pf_conv1x1_split3(const pf_conv_params* p, const void* w3, int w3_rows, void* stream) {
  if (!p || !p->x || !p->y || !w3) return PF_ERR_ARG;
#ifdef PF_C1_DBG
  pf_conv_params pd = *p;
  if (const char* s = getenv("PF_C1_DBG")) pd.pad = atoi(s);
  p = &pd;
  if (p->dtype != PF_DTYPE_F32 || p->KH != 1 || p->KW != 1 || p->stride != 1 || p->shuffle > 1) return PF_ERR_ARG;
#else
  if (p->dtype != PF_DTYPE_F32 || p->KH != 1 || p->KW != 1 || p->stride != 1 || p->pad != 0 || p->shuffle > 1) return PF_ERR_ARG;
#endif
  if (p->Cin <= 0 || p->Cin % 32 || p->x_ld % 4 || p->x_ld < p->Cin) return PF_ERR_ARG;
  if (p->Cout <= 0 || p->Cout % 4 || p->y_ld % 4 || w3_rows < p->Cout || w3_rows % 16) return PF_ERR_ARG;
  if ((p->res && p->res_ld % 4) || (p->res2 && p->res2_ld % 4)) return PF_ERR_ARG;
  const long M = (long)p->B * p->OH * p->OW;
  if (M <= 0 || M >= (1L << 31) || (long)w3_rows * 64 >= (1L << 31)) return PF_ERR_ARG;
  if ((reinterpret_cast<size_t>(p->x) | reinterpret_cast<size_t>(p->y) | reinterpret_cast<size_t>(w3)) & 15) return PF_ERR_ARG;
  // PF_C1_BM = 64 | 128 forces a token tile (A/B, tests); default 64 (two blocks per CU).  PF_C1_PERSIST=0: one tile per block (A/B, tests).
  static const int force = [] { const char* e = getenv("PF_C1_BM"); return e ? atoi(e) : 0; }();
  static const bool persist = [] { const char* e = getenv("PF_C1_PERSIST"); return !(e && e[0] == '0'); }();
  static const int cus = [] { int d = 0, n = 0; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
  const int bm = force == 128 ? 128 : 64;
  const int mt = (int)((M + bm - 1) / bm), nt = (p->Cout + BN - 1) / BN;
  if ((long)mt * nt >= (1L << 31)) return PF_ERR_ARG;
  // the grid: nslots token-tile slots per channel tile; persistent = as many blocks as are resident at once (two per CU at BM = 64, one at 128)
  int nslots = mt;
  if (persist) {
    const int resident = (bm == 64 ? 2 : 1) * cus;
    nslots = resident / nt > 0 ? resident / nt : 1;
    if (nslots > mt) nslots = mt;
  }
  if (const char* e = getenv("PF_C1_SLOTS")) {               // tests: a small grid makes every block cross many tile boundaries
    const int v = atoi(e);
    if (v > 0) nslots = v < mt ? v : mt;
  }
  constexpr int lds64 = 3 * 64 * 128 + 2 * WSTAGE, lds128 = 3 * 128 * 128 + 2 * WSTAGE;
  // (the attribute belongs to a device: one bit per device, as in gemm_split3.hip -- a process that drives several GPUs sets it on each)
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_split3_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_split3_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds128) != hipSuccess)
      return PF_ERR_LAUNCH;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const long w_bstride = (long)(p->Cin / 32) * w3_rows * 32;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* w = static_cast<const bf16_t*>(w3);
  const dim3 grid((unsigned)(nslots * nt));
  if (bm == 128) hipLaunchKernelGGL(conv1x1_split3_kernel<128>, grid, dim3(512), lds128, st, *p, w, w3_rows, w_bstride, mt, nt, nslots);
  else hipLaunchKernelGGL(conv1x1_split3_kernel<64>, grid, dim3(256), lds64, st, *p, w, w3_rows, w_bstride, mt, nt, nslots);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}
