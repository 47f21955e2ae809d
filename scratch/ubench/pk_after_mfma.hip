// Reproducer attempt for round 4's sporadic wrong rows (profiles/r04_nodeblock.txt item 9c): matrix-core result -> v_pk_add_f32 ->
// v_pk_fma_f32 ... op_sel:[0,1,0] (the low result takes the HIGH half of the packed sum), two workgroups per CU (two waves per SIMD),
// whole chip, many rounds.  The same source is compiled twice -- packed-fp32 on (KERNEL=k_packed) and off (KERNEL=k_scalar) -- and the
// host compares the two outputs bit for bit and each against itself over repeated launches.
//   hipcc -O3 --offload-arch=gfx950 -DBUILD_PACKED -c pk_after_mfma.hip -o pk_p.o
//   hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -c pk_after_mfma.hip -o pk_s.o   (+ -DBUILD_MAIN)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Args { const float* in; float* out; int rounds; };
#ifdef BUILD_PACKED
#define KERNEL k_packed
#else
#define KERNEL k_scalar
#endif
__global__ void __launch_bounds__(256, 2) KERNEL(Args a) {
  extern __shared__ float lds[];                       // 72 KB requested: exactly two workgroups fit a CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* p = a.in + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  bf16x8 w, x;
  f32x4 u, c;
  for (int j = 0; j < 8; ++j) { w[j] = (__bf16)p[j]; x[j] = (__bf16)p[8 + j]; }
  for (int j = 0; j < 4; ++j) { u[j] = p[j] * 0.37f; c[j] = p[4 + j] - 0.11f; }
  lds[threadIdx.x] = p[0];
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, res = {0.f, 0.f, 0.f, 0.f}, res2 = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < a.rounds; ++r) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);      // matrix-core result ...
    const f32x4 s = acc + c;                                                // ... -> v_pk_add_f32 (x2)
    // pairs of results scaled by ONE half of a packed sum: v_pk_fma_f32 v[lo:hi], u[lo:hi], s[pair], res[lo:hi] with op_sel:[0,1,0]
    // (both lanes take the HIGH half) or op_sel_hi:[1,0,1] (both take the LOW half)
    res[0] = __builtin_fmaf(u[0], s[1], res[0]);
    res[1] = __builtin_fmaf(u[1], s[1], res[1]);
    res[2] = __builtin_fmaf(u[2], s[0], res[2]);
    res[3] = __builtin_fmaf(u[3], s[0], res[3]);
    res2[0] = __builtin_fmaf(u[0], s[3], res2[0]);
    res2[1] = __builtin_fmaf(u[1], s[3], res2[1]);
    res2[2] = __builtin_fmaf(u[2], s[2], res2[2]);
    res2[3] = __builtin_fmaf(u[3], s[2], res2[3]);
    acc *= 0.25f;                                                           // keeps the sums bounded
    x[r & 7] = (__bf16)(lds[(threadIdx.x + 17 * r) & 255] * 0.01f);
  }
  res += res2;
  float* o = a.out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (int j = 0; j < 4; ++j) o[j] = res[j] + (wave == 9 ? lds[lane] : 0.f);
}
#ifdef BUILD_MAIN
__global__ void k_packed(Args a);
int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 2048, rounds = argc > 2 ? atoi(argv[2]) : 2000, reps = argc > 3 ? atoi(argv[3]) : 20;
  const size_t n = (size_t)wgs * 256;
  float* h = (float*)malloc(n * 16 * 4);
  srand(1);
  for (size_t i = 0; i < n * 16; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  float *din, *dout;
  hipMalloc(&din, n * 64); hipMalloc(&dout, n * 16);
  hipMemcpy(din, h, n * 64, hipMemcpyHostToDevice);
  float* first[2] = {(float*)malloc(n * 16), (float*)malloc(n * 16)};
  float* got = (float*)malloc(n * 16);
  Args a{din, dout, rounds};
  long self_bad[2] = {0, 0};
  for (int rep = 0; rep < reps; ++rep)
    for (int k = 0; k < 2; ++k) {
      hipFuncSetAttribute((const void*)(k ? k_packed : k_scalar), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
      if (k) hipLaunchKernelGGL(k_packed, dim3(wgs), dim3(256), 72 * 1024, 0, a); else hipLaunchKernelGGL(k_scalar, dim3(wgs), dim3(256), 72 * 1024, 0, a);
      hipMemcpy(got, dout, n * 16, hipMemcpyDeviceToHost);
      if (rep == 0) memcpy(first[k], got, n * 16);
      else for (size_t i = 0; i < n * 4; ++i) self_bad[k] += memcmp(&got[i], &first[k][i], 4) != 0;
    }
  long cross = 0;
  for (size_t i = 0; i < n * 4; ++i) cross += memcmp(&first[0][i], &first[1][i], 4) != 0;
  printf("workgroups %d rounds %d reps %d: scalar build differs from its first launch in %ld values, packed build in %ld; packed vs scalar (first launches) %ld of %zu\n",
         wgs, rounds, reps, self_bad[0], self_bad[1], cross, n * 4);
  return 0;
}
#endif
