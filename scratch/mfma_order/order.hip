// Do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 round the same way?  Same k order, same inputs: compare bits of D = A B over K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 224;
__global__ void k32(const float* A, const float* B, float* D) {   // A[32][K], B[K][32], D[32][32]
  const int l = threadIdx.x, i = l & 31, kh = l >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + kh], B[(k + kh) * 32 + i], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = acc[r];
}
__global__ void k16(const float* A, const float* B, float* D) {   // one wave per 16 x 16 block
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  const int bi = blockIdx.x >> 1, bj = blockIdx.x & 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * bi + i) * K + k + kq], B[(k + kq) * 32 + 16 * bj + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(16 * bi + 4 * kq + r) * 32 + 16 * bj + i] = acc[r];
}
int main() {
  float *A, *B, *D1, *D2;
  hipMallocManaged(&A, 32 * K * 4); hipMallocManaged(&B, K * 32 * 4); hipMallocManaged(&D1, 4096); hipMallocManaged(&D2, 4096);
  int total_diff = 0;
  for (int trial = 0; trial < 50; ++trial) {
    srand(trial);
    for (int x = 0; x < 32 * K; ++x) { A[x] = (rand() / (float)RAND_MAX - 0.5f) * (trial % 3 == 0 ? 100.f : 1.f); B[x] = rand() / (float)RAND_MAX - 0.5f; }
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, A, B, D1);
    hipLaunchKernelGGL(k16, dim3(4), dim3(64), 0, 0, A, B, D2);
    hipDeviceSynchronize();
    int diff = 0, seq_fma = 0, seq_pair = 0;
    for (int x = 0; x < 1024; ++x) {
      diff += memcmp(&D1[x], &D2[x], 4) != 0;
      const int i = x / 32, j = x % 32;
      float s = 0.f;
      for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 32 + j], s);
      seq_fma += memcmp(&s, &D1[x], 4) != 0;
    }
    total_diff += diff;
    if (trial < 5) printf("trial %d: 32x32x2 vs 16x16x4 differ in %d of 1024; 32x32x2 vs sequential fmaf chain differ in %d\n", trial, diff, seq_fma);
  }
  printf("TOTAL differing outputs over 50 trials: %d\n", total_diff);
  return 0;
}
