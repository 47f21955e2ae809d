// Micro-benchmark: issue rate of v_fma_f32 / v_pk_fma_f32 with VGPR / SGPR operands, CH independent chains,
// at W waves per SIMD.  Prints cycles per wave-instruction per SIMD (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CH, bool SG>
__global__ void __launch_bounds__(256) k_fma(const float* __restrict__ p, float* out, int iters) {
  float acc[CH];
  float w = p[threadIdx.x];
  float s = SG ? p[blockIdx.x & 1] : p[threadIdx.x + 1];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = (float)c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rc = 0; rc < CH * 16; ++rc) {
      const int c = rc % CH;
      if (SG) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c]) : "s"(s), "v"(w));
      else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c]) : "v"(s), "v"(w));
    }
  }
  float t = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) t += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int CH, bool SG>
__global__ void __launch_bounds__(256) k_pk(const float* __restrict__ p, float* out, int iters) {
  f2 acc[CH];
  f2 w = {p[threadIdx.x], p[threadIdx.x + 2]};
  f2 sv = {p[threadIdx.x + 1], p[threadIdx.x + 3]};
  f2 ss = {p[blockIdx.x & 1], p[(blockIdx.x & 1) + 1]};
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = f2{(float)c, 1.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rc = 0; rc < CH * 16; ++rc) {
      const int c = rc % CH;
      if (SG) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(w), "s"(ss));
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(w), "v"(sv));
    }
  }
  float t = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) t += acc[c].x + acc[c].y;
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <typename K>
void run(const char* name, K kern, int ch, int wps, const float* p, float* out) {
  const int iters = 16384;
  const int blocks = 256 * wps;  // 256-thread block = 1 wave per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, p, out, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, p, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)iters * 16 * ch * wps;
  printf("%-14s chains %2d waves/SIMD %d : %7.1f us  %.2f cycles/instr/SIMD\n", name, ch, wps, ms * 1e3, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float *p, *out;
  hipMalloc(&p, 4096 * 4); hipMemset(p, 0, 4096 * 4);
  hipMalloc(&out, 256 * 256 * 16 * 4);
  for (int wps : {1, 2, 4, 8}) {
    run("fma vgpr", k_fma<1, false>, 1, wps, p, out);
    run("fma vgpr", k_fma<4, false>, 4, wps, p, out);
    run("fma sgpr", k_fma<1, true>, 1, wps, p, out);
    run("fma sgpr", k_fma<4, true>, 4, wps, p, out);
    run("pk  vgpr", k_pk<1, false>, 1, wps, p, out);
    run("pk  vgpr", k_pk<4, false>, 4, wps, p, out);
    run("pk  sgpr", k_pk<1, true>, 1, wps, p, out);
    run("pk  sgpr", k_pk<4, true>, 4, wps, p, out);
  }
  return 0;
}
