// micro-benchmark 3: a prefetch stream (load issued one iteration ahead of its use, as the weight ring does) with and without
// row-per-lane stores in between: does the in-order vmcnt make the stream wait for the stores?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int REP = 64;
template <int MFMA, int NST, bool ROWS>
__global__ void __launch_bounds__(256, 1) k(const uint4* wsrc, float* out, unsigned long long* cyc, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wblk = (int64_t)blockIdx.x * 4 + wave;
  float* p = out + wblk * 65536 + (ROWS ? (int64_t)(lane & 31) * 2048 + (lane >> 5) * 4 : (int64_t)lane * 4);
  const uint4* q = wsrc + wave * 192 + lane;
  f32x16 acc = {0};
  bf16x8 b = {1, 1, 1, 1, 1, 1, 1, 1};
  float4 v = make_float4(lane, wave, 1.f, 2.f);
  uint4 pf0 = q[0], pf1 = q[64], pf2 = q[128];
  q += 768;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
    // use the fetched piece (forces the wait), issue the next fetch, then products with stores between them
    bf16x8 a0 = __builtin_bit_cast(bf16x8, pf0), a1 = __builtin_bit_cast(bf16x8, pf1), a2 = __builtin_bit_cast(bf16x8, pf2);
#pragma unroll
    for (int m = 0; m < NST; ++m) {
      *reinterpret_cast<float4*>(p) = v;
      p += ROWS ? 8 : 256;
    }
    __builtin_amdgcn_sched_barrier(0);
    pf0 = q[0]; pf1 = q[64]; pf2 = q[128];
    q += 768;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MFMA; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m % 3 == 0 ? a0 : (m % 3 == 1 ? a1 : a2), b, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[wblk] = t1 - t0;
  if (acc[0] == 12345.f) sink[0] = acc[1] + pf0.x + pf1.x + pf2.x;
}
template <int MFMA, int NST, bool ROWS>
static double run(int wgs, const uint4* w, float* out, unsigned long long* cyc, float* sink) {
  std::vector<unsigned long long> h(wgs * 4);
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL((k<MFMA, NST, ROWS>), dim3(wgs), dim3(256), 0, 0, w, out, cyc, sink);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[h.size() / 2] / REP;
}
int main() {
  float* out; unsigned long long* cyc; float* sink; uint4* w;
  size_t floats = (size_t)256 * 4 * 65536 + (1 << 20);
  hipMalloc(&out, floats * 4); hipMalloc(&cyc, 256 * 4 * 8); hipMalloc(&sink, 64); hipMalloc(&w, 768 * 16 * (REP + 2));
  hipMemset(out, 0, floats * 4); hipMemset(w, 0, 768 * 16 * (REP + 2));
  for (int wgs : {1, 12, 146}) {
    printf("---- %d workgroups; STORES BEFORE THE FETCH; cycles per iteration = fetch of 3 KB/wave one iteration ahead + 24 products (768 cyc) + stores\n", wgs);
    printf("no stores %7.1f | 1 row store %7.1f  2: %7.1f  4: %7.1f  8: %7.1f | contiguous 1: %7.1f  4: %7.1f  8: %7.1f\n", run<24, 0, true>(wgs, w, out, cyc, sink),
           run<24, 1, true>(wgs, w, out, cyc, sink), run<24, 2, true>(wgs, w, out, cyc, sink), run<24, 4, true>(wgs, w, out, cyc, sink), run<24, 8, true>(wgs, w, out, cyc, sink),
           run<24, 1, false>(wgs, w, out, cyc, sink), run<24, 4, false>(wgs, w, out, cyc, sink), run<24, 8, false>(wgs, w, out, cyc, sink));
  }
  return 0;
}
