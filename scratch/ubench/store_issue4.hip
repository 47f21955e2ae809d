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
  const int lpr = (int)half_off;
  float* p = out + wblk * 65536 + (int64_t)(lane / lpr) * lane_stride + (lane % lpr) * 4;
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
  for (int wgs : {1, 146}) {
    printf("---- %d workgroups of 4 waves: dwordx4 store, lanes per row (16 B each), row stride 576 floats; cycles per instruction per wave\n", wgs);
    for (int lpr : {1, 2, 4, 8, 16, 32, 64}) {
      // rows used per instruction: 64 / lpr; successive instructions move to the next group of rows (step = rows * stride) -- stays inside 65536 floats for REP = 64? rows*576*64 = up to 2.3M: no -> step along the row instead
      const int64_t step = lpr * 4;   // next 16 B * lpr along the same rows (576-float rows hold 144 / lpr steps; REP = 64 needs lpr <= 2 ... wrap not needed for timing: addresses run into the next rows)
      printf("lanes/row %2d (%2d rows x %4d B): mfma 0 %6.1f   mfma 6 %6.1f   mfma 24 %6.1f\n", lpr, 64 / lpr, lpr * 16, run<0, 4>(wgs, out, 576, lpr, step, cyc, sink),
             run<6, 4>(wgs, out, 576, lpr, step, cyc, sink), run<24, 4>(wgs, out, 576, lpr, step, cyc, sink));
    }
  }
  return 0;
}
