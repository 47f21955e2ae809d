// f32 MFMA issue-rate probe: NACC independent accumulators, back-to-back, W waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(1024) probe(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = (float)(threadIdx.x + r + a);
  float x = out[threadIdx.x & 63], y = out[64 + (threadIdx.x & 63)];
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[128 + blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC> void run(int waves_per_block, int blocks, int iters) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (128 + blocks * 1024) * 4 + 4096); hipMemset(out, 0, 1024); hipMalloc(&cyc, blocks * 16 * 8);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double n_mfma = (double)iters * 8 * NACC;
  printf("NACC %d waves/block %2d blocks %4d: %.1f cycles per MFMA per wave (wave 0), kernel %.1f us -> %.1f TF\n", NACC, waves_per_block, blocks,
         h[0] / n_mfma, ms * 1e3, n_mfma * waves_per_block * blocks * 4096 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(4, 1, 200); run<2>(4, 1, 100); run<4>(4, 1, 50);
  run<2>(8, 1, 100); run<2>(4, 256, 100); run<2>(8, 256, 100); run<2>(12, 256, 100); run<4>(4, 256, 200); run<4>(4, 1024, 200);
  return 0;
}
