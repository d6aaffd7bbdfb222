// Measurement-only (profiles/r6_issue_probe.md): do the MFMAs of one wave block the VALU / LDS / store issue of the OTHER wave on its SIMD?
// One 512-thread block per CU: waves 0-3 ("A", one per SIMD) run a pure MFMA stream, waves 4-7 ("B", the second wave of each SIMD) run a stream of plain VALU
// (or ds_read_b128, or global stores).  Cycles (s_memtime, wave 0 / wave 4 of block 0) for: A alone, B alone, both together -- 16x16x32 and 32x32x16.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe tools/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE, int KIND>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters, int run_a, int run_b) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  float s = 0;
  long long t0 = 0, t1 = 0;
  if (wave < 4) {
    if (run_a) {
      f32x4 acc[4] = {};
      f32x16 big[2] = {};
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x % 7 + i); b[i] = (__bf16)(float)(threadIdx.x % 5 - i); }
      t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u & 3], 0, 0, 0);
          else if (u & 1) big[(u >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[(u >> 1) & 1], 0, 0, 0);   // 8 x 32 cycles = 16 x 16 cycles
        }
      }
      t1 = __builtin_amdgcn_s_memtime();
      for (int i = 0; i < 4; ++i) s += acc[i][0];
      s += big[0][0] + big[1][3];
    }
  } else if (run_b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const float c0 = 1.0001f, c1 = 0.5f;
    float4 v = make_float4(0, 0, 0, 0);
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u & 7]) : "v"(c0), "v"(c1));
        else if (KIND == 1) { f32x4 w; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((unsigned)((((threadIdx.x & 63) * 4 + u * 256) & 4095) * 4))); v.x += w[0]; }
        else out[(size_t)(blockIdx.x * 512 + threadIdx.x) * 4 + (u & 3) + 1048576] = x[u & 7];
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) s += x[i];
    s += v.x;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}
template <int MODE, int KIND>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, (size_t)(1048576 + 256 * 512 * 4 + 16) * sizeof(float)); hipMalloc(&cyc, 16);
  const int iters = 1000;
  long long h[3][2];
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int ra = cfg != 1, rb = cfg != 0;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, ra, rb);
    hipMemcpy(h[cfg], cyc, 16, hipMemcpyDeviceToHost);
  }
  printf("%-44s A alone %8.1f | B alone %8.1f | together: A %8.1f  B %8.1f   (cycles per 16 MFMA-slots of 16 cycles / per 32 B-instructions)\n", name, h[0][0] / (double)iters,
         h[1][1] / (double)iters, h[2][0] / (double)iters, h[2][1] / (double)iters);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 0>("A: 16x16x32 stream, B: v_fma_f32");
  run<1, 0>("A: 32x32x16 stream, B: v_fma_f32");
  run<0, 1>("A: 16x16x32 stream, B: ds_read_b128");
  run<1, 1>("A: 32x32x16 stream, B: ds_read_b128");
  run<0, 2>("A: 16x16x32 stream, B: global_store_dword");
  run<1, 2>("A: 32x32x16 stream, B: global_store_dword");
  return 0;
}
