// micro-benchmark 2: cost of one global store instruction to a wave alone on its SIMD, vs. width, shape, density and number of workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int REP = 64;
template <int MFMA, int W>   // W dwords per lane
__global__ void __launch_bounds__(256, 1) k_store(float* out, int64_t lane_stride, int64_t half_off, int64_t step, unsigned long long* cyc, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wblk = (int64_t)blockIdx.x * 4 + wave;
  float* p = out + wblk * 65536 + (int64_t)(lane & 31) * lane_stride + (lane >> 5) * half_off;
  f32x16 acc = {0};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  float4 v = make_float4(lane, wave, 1.f, 2.f);
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
    if (W == 4) *reinterpret_cast<float4*>(p) = v;
    if (W == 2) *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
    if (W == 1) *p = v.x;
    p += step;
#pragma unroll
    for (int m = 0; m < MFMA; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[wblk] = t1 - t0;
  if (acc[0] == 12345.f) sink[0] = acc[1];
}
template <int MFMA, int W>
static double run(int wgs, float* out, int64_t ls, int64_t ho, int64_t step, unsigned long long* cyc, float* sink) {
  std::vector<unsigned long long> h(wgs * 4);
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL((k_store<MFMA, W>), dim3(wgs), dim3(256), 0, 0, out, ls, ho, step, cyc, sink);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[h.size() / 2] / REP;
}
int main() {
  float* out; unsigned long long* cyc; float* sink;
  size_t floats = (size_t)256 * 4 * 65536 + (1 << 20);
  hipMalloc(&out, floats * 4); hipMalloc(&cyc, 256 * 4 * 8); hipMalloc(&sink, 64);
  hipMemset(out, 0, floats * 4);
  for (int wgs : {1, 12, 146, 256}) {
    printf("---- %d workgroups of 4 waves\n", wgs);
    // row-per-lane: lane n -> row n (stride 2048 floats), halves 4 floats apart, step 8 floats ; contiguous: lane stride W, half off 32 W, step 64 W
#define ROW(M, W) run<M, W>(wgs, out, 2048, W, 2 * W, cyc, sink)
#define CON(M, W) run<M, W>(wgs, out, W, 32 * W, 64 * W, cyc, sink)
    printf("mfma 0 : row-per-lane x4 %6.1f x2 %6.1f x1 %6.1f | contiguous x4 %6.1f x2 %6.1f x1 %6.1f\n", ROW(0, 4), ROW(0, 2), ROW(0, 1), CON(0, 4), CON(0, 2), CON(0, 1));
    printf("mfma 6 (192 cyc): row-per-lane x4 %6.1f x2 %6.1f x1 %6.1f | contiguous x4 %6.1f x2 %6.1f x1 %6.1f\n", ROW(6, 4), ROW(6, 2), ROW(6, 1), CON(6, 4), CON(6, 2), CON(6, 1));
    printf("mfma 24 (768 cyc): row-per-lane x4 %6.1f x2 %6.1f x1 %6.1f | contiguous x4 %6.1f x2 %6.1f x1 %6.1f\n", ROW(24, 4), ROW(24, 2), ROW(24, 1), CON(24, 4), CON(24, 2), CON(24, 1));
  }
  return 0;
}
