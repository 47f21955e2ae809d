// memory-only probes of the message gather pattern (no filter, no MFMA)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: thread = channel (256 thr/block), block walks nodes, per edge: 3 dword loads of h + 1 of xhat (row 1056 floats)
extern "C" __global__ void __launch_bounds__(256) probe_chan(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ nbr,
                                                             const float* __restrict__ rows, int64_t N, int W, float* __restrict__ out) {
  const int t = threadIdx.x;
  for (int64_t c = blockIdx.x; c < N; c += gridDim.x) {
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const int e0 = rowptr[c], e1 = rowptr[c + 1];
#pragma unroll 4
    for (int e = e0; e < e1; ++e) {
      const float* r = rows + nbr[e] * W;
      acc0 += r[t]; acc1 += r[256 + t]; acc2 += r[512 + t]; acc3 += r[768 + t];
    }
    out[c * 256 + t] = acc0 + acc1 + acc2 + acc3;
  }
}
// B: wave per node, lane = float4 piece: each edge row (W floats) read as W/256 float4 per lane
extern "C" __global__ void __launch_bounds__(256) probe_wave(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ nbr,
                                                             const float* __restrict__ rows, int64_t N, int W, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t c = wid; c < N; c += nw) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const int e0 = rowptr[c], e1 = rowptr[c + 1];
#pragma unroll 2
    for (int e = e0; e < e1; ++e) {
      const f32x4* r = reinterpret_cast<const f32x4*>(rows + nbr[e] * W);
      a0 += r[lane]; a1 += r[64 + lane]; a2 += r[128 + lane]; a3 += r[192 + lane];
    }
    f32x4 s = a0 + a1 + a2 + a3;
    reinterpret_cast<f32x4*>(out + c * 256)[lane] = s;
  }
}

// C: probe_chan + the 60 filter FMAs fed by a scalar (workgroup-uniform) 32-float record per edge
extern "C" __global__ void __launch_bounds__(256) probe_fma(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ nbr,
                                                            const float* __restrict__ rows, int64_t N, int W, float* __restrict__ out,
                                                            const float* __restrict__ eb, const float* __restrict__ wts) {
  const int t = threadIdx.x;
  float w0[20], w1[20], w2[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) { w0[k] = wts[t * 20 + k]; w1[k] = wts[(256 + t) * 20 + k]; w2[k] = wts[(512 + t) * 20 + k]; }
  for (int64_t c = blockIdx.x; c < N; c += gridDim.x) {
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const int e0 = rowptr[c], e1 = rowptr[c + 1];
#pragma unroll 4
    for (int e = e0; e < e1; ++e) {
      const float* r = rows + nbr[e] * W;
      const float* rec = eb + (int64_t)e * 32;
      float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int k = 0; k < 20; ++k) { p0 += w0[k] * rec[k]; p1 += w1[k] * rec[k]; p2 += w2[k] * rec[k]; }
      acc0 += r[t] * p0; acc1 += r[256 + t] * p1; acc2 += r[512 + t] * p2; acc3 += r[768 + t] * (p0 + rec[21]);
    }
    out[c * 256 + t] = acc0 + acc1 + acc2 + acc3;
  }
}
