// Measurement-only: what ds_read_b64_tr_b16 returns.  Every lane of a wave points at 4 consecutive 16-bit elements of an LDS array whose element i holds
// the value i; the program prints, for each of the first 20 lanes and lanes 32..35, the four element indices it received.  Expected (and what
// csrc/attn_split3.hip assumes): inside a 16-lane group, lane i receives element (i & 3) of the four lanes 4 e + (i >> 2), e = 0..3 -- with the lanes of
// the group pointing at a row-major [4][16] block (lane j at row j >> 2, columns 4 (j & 3) ..), lane i receives column i.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/tr_probe tools/tr_probe.hip ; run on the GPU box: ./tools/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) v4s lds_v4s;
__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short a[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) a[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, j = lane & 15, grp = lane >> 4;
  // group grp reads rows 4 grp .. 4 grp + 3 of a [16][64] array (128-byte rows), columns 0..15
  const short* p = a + (4 * grp + (j >> 2)) * 64 + 4 * (j & 3);
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)p);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d;
  hipMalloc(&d, 256 * sizeof(short));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  short h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 4; ++e) ok &= h[lane * 4 + e] == (4 * (lane >> 4) + e) * 64 + (lane & 15);
  for (int lane = 0; lane < 36; ++lane)
    if (lane < 20 || lane >= 32) printf("lane %2d: %4d %4d %4d %4d   (row, col) = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2],
                                        h[lane * 4 + 3], h[lane * 4] / 64, h[lane * 4] % 64, h[lane * 4 + 1] / 64, h[lane * 4 + 1] % 64, h[lane * 4 + 2] / 64,
                                        h[lane * 4 + 2] % 64, h[lane * 4 + 3] / 64, h[lane * 4 + 3] % 64);
  printf("column-of-a-[4][16]-block semantic: %s\n", ok ? "CONFIRMED" : "NOT what the kernel assumes");
  return 0;
}
