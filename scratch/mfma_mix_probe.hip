// What keeps the f32 matrix pipe from 100 % when MFMAs are mixed with the loads of a GEMM body?  Per wave and iteration: LD global
// 16-byte loads (fragment-order weights, L2-resident), DS ds_read_b128 per 8 MFMAs, 32 MFMAs on two accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LD, int DS, bool SB>
__global__ void __launch_bounds__(256) probe(const float4* __restrict__ w, float* out, int iters, int wstride) {
  __shared__ __attribute__((aligned(16))) float lds[32 * 132];
  for (int i = threadIdx.x; i < 32 * 132; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  const float4* wp = w + (size_t)(blockIdx.x % 64) * wstride + wave * 4096 + lane;
  float4 b[8];
  for (int q = 0; q < 8; ++q) b[q] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* ts = lds + (lane & 31) * 132 + 4 * (lane >> 5);
  for (int it = 0; it < iters; ++it) {
    if (LD) {
#pragma unroll
      for (int q = 0; q < LD; ++q) b[q] = wp[((it * 8 + q) & 63) * 64];
    }
    if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 t = make_float4(1.f, 1.f, 1.f, 1.f);
      if (DS) t = *reinterpret_cast<const float4*>(ts + 8 * ((it + q) & 15));
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, t.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].x, t.x, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, t.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].y, t.y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, t.z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].z, t.z, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, t.w, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].w, t.w, a1, 0, 0, 0);
    }
    if (SB) __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int LD, int DS, bool SB> void run(const char* name, const float4* w, float* out, int blocks, int iters) {
  hipLaunchKernelGGL((probe<LD, DS, SB>), dim3(blocks), dim3(256), 0, 0, w, out, iters, 16384);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<LD, DS, SB>), dim3(blocks), dim3(256), 0, 0, w, out, iters, 16384);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 32 * blocks / 256.0;   // one wave per SIMD per block, blocks / 256 blocks per CU
  printf("%-44s blocks %5d: %.1f us, %.1f ns per MFMA per SIMD (64 cycles = %.1f ns at 2.4 GHz)\n", name, blocks, ms * 1e3, ms * 1e6 / mfma_per_simd, 64 / 2.4);
}
int main() {
  float4* w; float* out;
  (void)hipMalloc(&w, 64 * 16384 * 16 + (1 << 20)); (void)hipMemset(w, 0, 64 * 16384 * 16 + (1 << 20)); (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 400;
  for (int blocks : {256, 768}) {
    run<0, 0, false>("MFMA only", w, out, blocks, iters);
    run<0, 1, false>("MFMA + ds_read_b128 per 8", w, out, blocks, iters);
    run<8, 0, false>("MFMA + 8 global b128 per 32", w, out, blocks, iters);
    run<8, 1, false>("MFMA + both", w, out, blocks, iters);
    run<8, 1, true>("MFMA + both, loads fenced ahead", w, out, blocks, iters);
    run<4, 1, false>("MFMA + 4 global b128 per 32 + ds", w, out, blocks, iters);
  }
  return 0;
}
