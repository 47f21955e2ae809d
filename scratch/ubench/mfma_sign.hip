// Is v_mfma_f32_16x16x32_bf16 odd in one operand and its accumulator, bit for bit?  D = A B + C against -((-A) B + (-C)).
// (The node block's reverse launch is linear in its cotangents but g -> -g does not negate its result bit for bit: scratch/nb_sign.py.)
// Also: v_mfma_f32_32x32x16_bf16 (the message kernels' filter) and the exact-f32 v_mfma_f32_32x32x2_f32.
// build: hipcc -O2 --offload-arch=gfx950 mfma_sign.hip -o mfma_sign.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ float rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
__global__ void k(int* out, int iters) {
  uint32_t s = 12345u + threadIdx.x * 7919u + blockIdx.x * 104729u;
  int bad16 = 0, bad32 = 0, badf = 0;
  for (int it = 0; it < iters; ++it) {
    bf16x8 a, b, na;
    f32x4 c, nc;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(rnd(s) * 3.0f); b[i] = (__bf16)(rnd(s) * 3.0f); na[i] = (__bf16)(-(float)a[i]); }
    for (int i = 0; i < 4; ++i) { c[i] = rnd(s) * 40.0f; nc[i] = -c[i]; }
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(na, b, nc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) bad16 += (d[i] != -e[i]);
    f32x16 c2, nc2;
    for (int i = 0; i < 16; ++i) { c2[i] = rnd(s) * 40.0f; nc2[i] = -c2[i]; }
    const f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    const f32x16 e2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na, b, nc2, 0, 0, 0);
    for (int i = 0; i < 16; ++i) bad32 += (d2[i] != -e2[i]);
    const float fa = rnd(s) * 3.0f, fb = rnd(s) * 3.0f;
    const f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
    const f32x16 e3 = __builtin_amdgcn_mfma_f32_32x32x2f32(-fa, fb, nc2, 0, 0, 0);
    for (int i = 0; i < 16; ++i) badf += (d3[i] != -e3[i]);
  }
  atomicAdd(&out[0], bad16);
  atomicAdd(&out[1], bad32);
  atomicAdd(&out[2], badf);
}
int main() {
  int* d;
  hipMalloc(&d, 12);
  hipMemset(d, 0, 12);
  const int blocks = 64, iters = 2000;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters);
  int h[3];
  hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  const long n = (long)blocks * 64 * iters;
  printf("results whose bits are not the negation of the negated problem's:\n");
  printf("  v_mfma_f32_16x16x32_bf16: %d of %ld\n  v_mfma_f32_32x32x16_bf16: %d of %ld\n  v_mfma_f32_32x32x2_f32:   %d of %ld\n", h[0], n * 4, h[1], n * 16, h[2], n * 16);
  return 0;
}
