// Micro-benchmark: cost of scalar loads (s_load_dwordx16 + x8 + x4 of a 128-B record) per wave, as the sb message
// kernels issue them: W waves per SIMD, each wave walks its own stream of records (stride = 128 B) or a shared one.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>  // 0: every wave of the block reads the same records (block-shared stream); 1: one record for all
__global__ void __launch_bounds__(256) k_smem(const float* __restrict__ rec, float* out, int n_rec, int iters) {
  float acc = 0.f;
  const float w = (float)threadIdx.x;
  size_t base = (size_t)blockIdx.x * iters;
  for (int i = 0; i < iters; ++i) {
    const size_t r = MODE == 1 ? 0 : (base + i) % n_rec;
    const float* p = rec + r * 32;  // uniform
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += w * p[k];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_vmem(const float* __restrict__ rec, float* out, int n_rec, int iters) {
  float acc = 0.f;
  const float w = (float)threadIdx.x;
  size_t base = (size_t)blockIdx.x * iters;
  const int z = threadIdx.x >> 10;  // 0, but not provably uniform: forces vector loads of a wave-uniform address
  for (int i = 0; i < iters; ++i) {
    const size_t r = MODE == 1 ? 0 : (base + i) % n_rec;
    const float4* p = reinterpret_cast<const float4*>(rec + r * 32) + z;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float4 v = p[k]; acc += w * (v.x + v.y + v.z + v.w); }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename K>
void run(const char* name, K kern, int bpc, const float* rec, float* out, int n_rec) {
  const int iters = 2048, blocks = 256 * bpc;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, rec, out, n_rec, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, rec, out, n_rec, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: bpc blocks x iters records x 4 waves
  printf("%-28s blocks/CU %d : %8.1f us  %6.1f ns per record per block  (%.0f cycles @2.4GHz)\n", name, bpc, ms * 1e3, ms * 1e6 / iters, ms * 1e-3 / iters * 2.4e9);
}
int main() {
  const int n_rec = 320000;
  float *rec, *out;
  hipMalloc(&rec, (size_t)n_rec * 128); hipMemset(rec, 0, (size_t)n_rec * 128);
  hipMalloc(&out, 256 * 256 * 8 * 4);
  for (int bpc : {1, 2, 4}) {
    run("smem stream (miss+3 hits)", k_smem<0>, bpc, rec, out, n_rec);
    run("smem same record (hits)", k_smem<1>, bpc, rec, out, n_rec);
    run("vmem-uniform stream", k_vmem<0>, bpc, rec, out, n_rec);
    run("vmem-uniform same record", k_vmem<1>, bpc, rec, out, n_rec);
  }
  return 0;
}
