// What does the gfx950 matrix pipe SUSTAIN?  (round-4 review item: "measure the MFMA ceiling instead of arguing it")
//
// Register-resident bf16 MFMA streams: no LDS, no DMA, no global traffic inside the timed loop.  Every wave holds the fragment set of one K chunk of
// the split-precision GEMM's 192 x 192 tile (csrc/gemm_split3.hip: 3 planes x (6 W + 3 X) fragments, 18 accumulators) and issues that chunk's 108
// v_mfma_f32_16x16x32_bf16 over and over (the six-term order of the product kernel), or the same FLOPs as v_mfma_f32_32x32x16_bf16.
//   data   zeros | random (three independent N(0,1) bf16 planes) | hml (the h / m / l planes of random float32 values: |m| ~ 2^-8 |h|, |l| ~ 2^-16 |h|)
//   sched  free      both waves of a SIMD issue MFMAs all the time (8 waves per CU, 2 per SIMD): the pipe is never idle
//          pingpong  the product kernel's barrier skeleton with everything else removed: waves 0-3 multiply in even phases, waves 4-7 in odd phases,
//                    one s_barrier between phases (a wave's "load phase" is empty) -- the floor the phase structure alone imposes
//          solo      one wave per SIMD (4 waves per CU), free running
// Per case: wall time of the launch (HIP events), executed TFLOP/s, the effective shader clock = s_memtime ticks / s_memrealtime (100 MHz) time of one
// block, and the pipe occupancy = MFMA issue cycles (16 per 16x16x32, 32 per 32x32x16, per SIMD) / elapsed shader cycles of that block.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_ceiling tools/mfma_ceiling.hip      run: tools/mfma_ceiling [ms per launch, default 40]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { SCHED_FREE = 0, SCHED_PINGPONG = 1 };

// frag: [9 fragments per lane-slot][64 lanes] uint4 per wave-slot (8 wave slots), the same for every block
template <int SCHED>
__global__ __launch_bounds__(512) void mfma16_kernel(const uint4* __restrict__ frag, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 w[3][6], x[3][3];
  const uint4* f = frag + (size_t)wave * 27 * 64 + lane;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
    for (int i = 0; i < 6; ++i) w[pl][i] = f[(pl * 9 + i) * 64];
#pragma unroll
    for (int i = 0; i < 3; ++i) x[pl][i] = f[(pl * 9 + 6 + i) * 64];
  }
  f32x4 acc[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define TERM(PW, PX)                                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 6; ++i) _Pragma("unroll") for (int j = 0; j < 3; ++j)                                  \
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[PW][i]), __builtin_bit_cast(bf16x8, x[PX][j]), acc[i][j], 0, 0, 0);
  const bool grp_b = wave >= 4;
  if (SCHED == SCHED_PINGPONG && grp_b) __builtin_amdgcn_s_barrier();           // group B runs one phase behind
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    if (SCHED == SCHED_PINGPONG) __builtin_amdgcn_s_barrier();                 // (the empty load phase ends)
    __builtin_amdgcn_sched_barrier(0);
    TERM(0, 2) TERM(2, 0) TERM(1, 1) TERM(0, 1) TERM(1, 0) TERM(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    if (SCHED == SCHED_PINGPONG) __builtin_amdgcn_s_barrier();                 // the compute phase ends
  }
  if (SCHED == SCHED_PINGPONG && !grp_b) __builtin_amdgcn_s_barrier();
#undef TERM
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) s += acc[i][j];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// the same FLOPs per iteration as 32x32x16 products: 54 MFMAs of 32768 FLOP each (3 W x 3 X fragments x 3 plane pairs, twice), 9 accumulators of 16 registers
__global__ __launch_bounds__(512) void mfma32_kernel(const uint4* __restrict__ frag, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 w[3][3], x[3][3];
  const uint4* f = frag + (size_t)wave * 27 * 64 + lane;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int i = 0; i < 3; ++i) { w[pl][i] = f[(pl * 9 + i) * 64]; x[pl][i] = f[(pl * 9 + 6 + i) * 64]; }
  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)                    // 54 MFMAs of 32768 FLOP = the 1.77 MFLOP of the 16x16x32 kernel's iteration
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[pl][i]), __builtin_bit_cast(bf16x8, x[rep ? pl : 2 - pl][j]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// fragment image: [8 waves][3 planes][9 fragments][64 lanes][8 bf16]
static std::vector<uint16_t> make_frags(const char* data) {
  const size_t n = (size_t)8 * 27 * 64 * 8;
  std::vector<uint16_t> v(n, 0);
  if (!strcmp(data, "zeros")) return v;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  const size_t per_plane = (size_t)9 * 64 * 8;
  for (int wv = 0; wv < 8; ++wv)
    for (size_t e = 0; e < per_plane; ++e) {
      uint16_t* p = v.data() + (size_t)wv * 3 * per_plane + e;
      if (!strcmp(data, "random")) {
        for (int pl = 0; pl < 3; ++pl) p[pl * per_plane] = f2bf(nd(rng));
      } else {                                        // hml: the exact split of a random float32
        const float xv = nd(rng);
        const uint16_t h = f2bf(xv);
        const float r1 = xv - bf2f(h);
        const uint16_t m = f2bf(r1);
        const uint16_t l = f2bf(r1 - bf2f(m));
        p[0] = h; p[per_plane] = m; p[2 * per_plane] = l;
      }
    }
  return v;
}

int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 40.0;
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  printf("# bf16 MFMA ceiling, register-resident streams (tools/mfma_ceiling.hip); %s, %d CUs, max clock %d MHz\n\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  printf("| instruction | data | schedule | waves/CU | ms/launch | executed TF/s | of 2500 | eff. clock GHz (block 0 / min / max) | pipe occupancy |\n");
  printf("|---|---|---|---:|---:|---:|---:|---|---:|\n");
  uint4* d_frag;
  float* d_out;
  unsigned long long* d_clk;
  CK(hipMalloc(&d_frag, (size_t)8 * 27 * 64 * 16));
  CK(hipMalloc(&d_out, (size_t)cus * 512 * 4));
  CK(hipMalloc(&d_clk, (size_t)cus * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Case { const char* inst; const char* data; const char* sched; int threads; };
  const Case cases[] = {
      {"16x16x32", "zeros", "free", 512},    {"16x16x32", "random", "free", 512},    {"16x16x32", "hml", "free", 512},
      {"16x16x32", "random", "pingpong", 512}, {"16x16x32", "hml", "pingpong", 512},  {"16x16x32", "zeros", "pingpong", 512},
      {"16x16x32", "random", "solo", 256},   {"16x16x32", "hml", "solo", 256},
      {"32x32x16", "zeros", "free", 512},    {"32x32x16", "random", "free", 512},    {"32x32x16", "hml", "free", 512},
      {"32x32x16", "random", "solo", 256},
      {"16x16x32", "random", "free", 512},   {"16x16x32", "hml", "pingpong", 512},   // (repeats: drift of the box over the run)
  };
  for (const Case& c : cases) {
    std::vector<uint16_t> h = make_frags(c.data);
    CK(hipMemcpy(d_frag, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const bool is16 = !strcmp(c.inst, "16x16x32");
    const bool pp = !strcmp(c.sched, "pingpong");
    const int waves = c.threads / 64;
    auto launch = [&](int iters) {
      if (!is16) hipLaunchKernelGGL(mfma32_kernel, dim3(cus), dim3(c.threads), 0, 0, d_frag, d_out, d_clk, iters);
      else if (pp) hipLaunchKernelGGL(mfma16_kernel<SCHED_PINGPONG>, dim3(cus), dim3(c.threads), 0, 0, d_frag, d_out, d_clk, iters);
      else hipLaunchKernelGGL(mfma16_kernel<SCHED_FREE>, dim3(cus), dim3(c.threads), 0, 0, d_frag, d_out, d_clk, iters);
    };
    // calibrate the iteration count for ~target_ms per launch, then 6 launches back to back (the clock settles over the first ones); report the last 4
    int iters = 2000;
    launch(iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    iters = (int)(iters * target_ms / (ms > 1e-3f ? ms : 1e-3f));
    if (iters < 100) iters = 100;
    launch(iters);
    launch(iters);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) launch(iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 4;
    std::vector<unsigned long long> clk((size_t)cus * 2);
    CK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    double gmin = 1e9, gmax = 0, g0 = 0, occ0 = 0;
    const double flop_iter_wave = 108.0 * 16384.0;                                     // both kernels: 1.77 MFLOP per wave and iteration
    const double mfma_cycles_iter = 108.0 * 16.0;                                      // pipe cycles per wave-iteration (27 x 32 x ... = the same 1728)
    for (int b = 0; b < cus; ++b) {
      const double ghz = (double)clk[2 * b] / ((double)clk[2 * b + 1] * 10.0);          // ticks / (realtime ticks x 10 ns)
      gmin = ghz < gmin ? ghz : gmin;
      gmax = ghz > gmax ? ghz : gmax;
      if (b == 0) { g0 = ghz; occ0 = (waves / 4.0) * iters * mfma_cycles_iter / (double)clk[0]; }
    }
    const double tf = (double)cus * waves * iters * flop_iter_wave / (ms * 1e-3) / 1e12;
    printf("| %s | %s | %s | %d | %.2f | %.0f | %.3f | %.2f / %.2f / %.2f | %.3f |\n", c.inst, c.data, c.sched, waves, ms, tf, tf / 2500.0, g0, gmin, gmax, occ0);
    fflush(stdout);
  }
  printf("\n(executed TF/s counts every MFMA issued; pipe occupancy = issue cycles of one SIMD's MFMAs / elapsed shader cycles of block 0, 1.0 = back to back)\n");
  return 0;
}
