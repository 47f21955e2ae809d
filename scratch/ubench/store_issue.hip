// micro-benchmark: what does one wave (alone on its SIMD) pay per global store / load instruction, by access shape?
// build: hipcc -O3 --offload-arch=gfx950 store_issue.hip -o store_issue ; run: ./store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int REP = 64;   // instructions per measurement
// shape: 0 row-per-lane (32 rows x 2 x 16 B, row stride `stride` floats), 1 contiguous 1 KB, 2: 8 rows x 128 B, 3: 4 rows x 256 B, 4: 16 rows x 64 B
__device__ __forceinline__ int64_t lane_off(int shape, int lane, int64_t stride) {
  switch (shape) {
    case 0: return (int64_t)(lane & 31) * stride + 4 * (lane >> 5);
    case 1: return 4 * lane;
    case 2: return (int64_t)(lane >> 3) * stride + 4 * (lane & 7);
    case 3: return (int64_t)(lane >> 4) * stride + 4 * (lane & 15);
    default: return (int64_t)(lane >> 2) * stride + 4 * (lane & 3);
  }
}
template <int MFMA>
__global__ void __launch_bounds__(256, 1) k_store(float* out, int64_t stride, int shape, int is_load, unsigned long long* cyc, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wblk = (int64_t)blockIdx.x * 4 + wave;
  // each wave owns 32 rows (shape 0) of `stride` floats; REP instructions walk the row in 32-float steps (8 floats.. as st_tile: 4 per tile)
  float* base = out + wblk * 32 * stride;
  const int64_t off = lane_off(shape, lane, stride);
  f32x16 acc = {0};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  float4 v = make_float4(lane, wave, 1.f, 2.f);
  float4 ld = make_float4(0, 0, 0, 0);
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
    // distinct addresses per instruction: shape 0 -> next 8 floats of the rows (as the 4 quads of st_tile walk 8 g + 4 h)
    int64_t step;
    switch (shape) {
      case 0: step = 8 * r; break;                 // 32 B per row per instruction
      case 1: step = 256 * r; break;               // next KB
      case 2: step = (r & 3) * 32 + (r >> 2) * 8 * stride; break;   // 128 B pieces: 4 per 8-row group then next 8 rows (wraps inside the wave's 32 rows x ...)
      case 3: step = (r & 1) * 64 + (r >> 1) * 4 * stride; break;
      default: step = (r & 7) * 16 + (r >> 3) * 16 * stride; break;
    }
    float4* p = reinterpret_cast<float4*>(base + off + step);
    if (is_load) {
      float4 t = *p;
      ld.x += t.x; ld.y += t.y; ld.z += t.z; ld.w += t.w;
    } else {
      *p = v;
    }
#pragma unroll
    for (int m = 0; m < MFMA; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[wblk] = t1 - t0;
  if (acc[0] + ld.x + ld.y == 12345.f) sink[0] = acc[1] + ld.z + ld.w;
}

int main() {
  const int wgs = 146;
  const int64_t stride_row = 576;   // h rows
  size_t floats = (size_t)wgs * 4 * 32 * 2048 + (1 << 20);
  float* out; unsigned long long* cyc; float* sink;
  hipMalloc(&out, floats * 4); hipMalloc(&cyc, wgs * 4 * 8); hipMalloc(&sink, 64);
  hipMemset(out, 0, floats * 4);
  std::vector<unsigned long long> h(wgs * 4);
  const char* names[] = {"row-per-lane 32x(2x16B)", "contiguous 1 KB", "8 rows x 128 B", "4 rows x 256 B", "16 rows x 64 B"};
  for (int is_load = 0; is_load < 2; ++is_load)
    for (int mf = 0; mf < 3; ++mf)
      for (int shape = 0; shape < 5; ++shape) {
        const int64_t stride = shape == 1 ? 2048 : (shape == 0 ? 2048 : 2048);   // a wave's 32 rows of 2048 floats hold every pattern
        for (int it = 0; it < 3; ++it) {
          if (mf == 0) hipLaunchKernelGGL(k_store<0>, dim3(wgs), dim3(256), 0, 0, out, stride, shape, is_load, cyc, sink);
          if (mf == 1) hipLaunchKernelGGL(k_store<6>, dim3(wgs), dim3(256), 0, 0, out, stride, shape, is_load, cyc, sink);
          if (mf == 2) hipLaunchKernelGGL(k_store<12>, dim3(wgs), dim3(256), 0, 0, out, stride, shape, is_load, cyc, sink);
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const int mfn = mf == 0 ? 0 : (mf == 1 ? 6 : 12);
        printf("%s  mfma/instr %2d  %-26s median %7.1f cycles per instruction (min %6.1f max %7.1f)  [mfma alone %d]\n", is_load ? "load " : "store", mfn, names[shape],
               (double)h[h.size() / 2] / REP, (double)h[0] / REP, (double)h.back() / REP, mfn * 32);
      }
  return 0;
}
