// Measurement-only (profiles/r6_issue_probe.md): how many plain VALU instructions fit in the shadow of one bf16 MFMA on gfx950, one and two waves per SIMD.
// Loop body = one MFMA (16x16x32, 16 pipe cycles; or 32x32x16, 32 pipe cycles) on rotating independent accumulators followed by K independent VALU
// instructions (v_fma_f32, or v_exp_f32 / v_cvt_pk_bf16_f32 mixes) written as asm volatile so the order is the source order.  Prints shader cycles per loop
// body from s_memtime (wave 0 of block 0) for K = 0..8.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe tools/issue_probe.hip ; run on the GPU box: ./tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int K, int MODE, int KIND>
__global__ __launch_bounds__(256, 2) void probe(float* out, long long* cyc, int iters) {
  f32x4 acc[4] = {};
  f32x16 big[2] = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x % 7 + i); b[i] = (__bf16)(float)(threadIdx.x % 5 - i); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  const float c0 = 1.0001f, c1 = 0.5f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u & 3], 0, 0, 0);
      else big[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[u & 1], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float& v = x[(u * K + k) & 7];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c0), "v"(c1));
        else if (KIND == 1) { if (k & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c0), "v"(c1)); }
        else { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(c0)); v = __uint_as_float(r << 16); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  s += big[0][0] + big[1][5];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int K, int MODE, int KIND>
double run(int blocks_per_cu) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 2 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<K, MODE, KIND>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((probe<K, MODE, KIND>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, cyc, iters);
  long long h = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  hipFree(out); hipFree(cyc);
  return (double)h / (iters * 8.0);
}
template <int MODE, int KIND>
void sweep(const char* name) {
  for (int bpc = 1; bpc <= 2; ++bpc) {
    printf("%-34s %d wave(s)/SIMD: cycles per {MFMA + K VALU}, K = 0..8:", name, bpc);
    printf(" %.1f", run<0, MODE, KIND>(bpc)); printf(" %.1f", run<1, MODE, KIND>(bpc)); printf(" %.1f", run<2, MODE, KIND>(bpc));
    printf(" %.1f", run<3, MODE, KIND>(bpc)); printf(" %.1f", run<4, MODE, KIND>(bpc)); printf(" %.1f", run<5, MODE, KIND>(bpc));
    printf(" %.1f", run<6, MODE, KIND>(bpc)); printf(" %.1f", run<8, MODE, KIND>(bpc));
    printf("\n");
  }
}
int main() {
  sweep<0, 0>("16x16x32 + v_fma_f32");
  sweep<0, 1>("16x16x32 + v_fma / v_exp mix");
  sweep<0, 2>("16x16x32 + v_cvt_pk_bf16 + shift");
  sweep<1, 0>("32x32x16 + v_fma_f32");
  sweep<1, 1>("32x32x16 + v_fma / v_exp mix");
  return 0;
}
