// Shared tile machinery of the fp32 MFMA message kernels (xeq_message_mfma.hip: general
// gather form; xeq_message_seg.hip: LDS-window form for closed node segments).
#pragma once
#include <stdlib.h>

#include "xeq_common.h"

namespace xeq {

constexpr int TE = 16;   // edges per tile (MFMA N)
constexpr int TPW = 9;   // channel tiles per wave: 4 waves x 9 x 16 = 576 channels max
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Msg2Args {
  int64_t n_nodes, n_edges;
  const int32_t* rowptr;
  const int32_t* perm;
  const int64_t* other_idx;  // node whose rows are gathered per edge (fwd: neighbor, bwd: center)
  int F, C, D, H, NT, HP;
  Irreps ir;
  RadialSpec rs;
  int xl;      // layout of xhat / grad_xhat (XAddr in xeq_common.h)
  int ablate;  // experiment switch (XEQ_ABLATE): 1 skip phase 0, 2 skip phase 1, 4 skip phase 2
};

#define XEQ_DPP_ADD(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xF, true)))

// sum over the 64 lanes; the total is valid in lane 63 only
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = XEQ_DPP_ADD(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
  v = XEQ_DPP_ADD(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
  v = XEQ_DPP_ADD(v, 0x141, 0xF);  // row_half_mirror
  v = XEQ_DPP_ADD(v, 0x140, 0xF);  // row_mirror          -> every lane holds its row's sum
  v = XEQ_DPP_ADD(v, 0x142, 0xA);  // row_bcast15 into rows 1, 3
  v = XEQ_DPP_ADD(v, 0x143, 0xC);  // row_bcast31 into rows 2, 3 -> lane 63 = total
  return v;
}

// first node n in [0, N] with rowptr[n] >= target
__device__ __forceinline__ int32_t lower_node(const int32_t* __restrict__ rowptr, int64_t N, int64_t target) {
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  return (int32_t)lo;
}

// node that owns CSR slot p: largest n in [n0, n1) with rowptr[n] <= p
__device__ __forceinline__ int32_t node_of_slot(const int32_t* __restrict__ rowptr, int64_t n0, int64_t n1, int32_t p) {
  int64_t lo = n0, hi = n1;  // first n in (n0, n1] with rowptr[n] > p
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid + 1] > p) hi = mid;
    else lo = mid + 1;
  }
  return (int32_t)lo;
}

template <int KS, bool BWD>
struct Smem {
  // rho/drho are dead once phase 1 has loaded its B fragments; the reverse pass reuses their
  // storage for the per-edge reduction slots of phase 2 (keeps the block under 80 KB of LDS)
  union {
    struct {
      float rho[TE][4 * KS + 1];
      float drho[BWD ? TE : 1][4 * KS + 1];
    };
    float red[BWD ? TE : 1][4][9];
  };
  float y[TE][12];     // [1 | Y1(3) | Y2(5) | f | f' | pad]
  float g[TE][5];      // unit vector, |r|, 1/|r|
  int32_t self[TE];
  int32_t other[TE];
  int32_t eid[TE];
  int32_t range[2];
};

// hardware sin/cos of an angle given in revolutions (v_sin_f32 / v_cos_f32 take S0 * 2 pi)
__device__ __forceinline__ void sincos_rev(float rev, float& s, float& c) {
  s = __builtin_amdgcn_sinf(rev);
  c = __builtin_amdgcn_cosf(rev);
}

// rho_k(d) and d rho_k / dd WITHOUT the envelope (applied in phase 1).  `w` = p0 / (2 pi) for the
// Bessel basis: the product d * w is range-reduced with its exact fma residual, so the argument
// error is that of v_sin_f32 itself, not of the fp32 product (the reference's torch.sin(freq * d)
// carries the product's rounding error, ~4e-6 rad at k = 20).
__device__ __forceinline__ void radial_fast(int kind, float d, float rc, float p0k, float p1k, float w, float& rho, float& drho) {
  if (kind == XEQ_RBF_BESSEL) {
    const float coeff = sqrtf(2.f / rc);
    const float p = d * w;
    const float res = fmaf(d, w, -p);
    float sn, cs;
    sincos_rev((p - floorf(p)) + res, sn, cs);
    const float inv = 1.f / (d + 1e-5f);
    rho = coeff * sn * inv;
    drho = coeff * (p0k * cs * inv - sn * inv * inv);
  } else {
    const float sd = fabsf(p1k) + 1e-5f;
    const float coeff = 1.f / (sd * 2.5066282746310002f);
    const float z = (d - p0k) / sd;
    rho = coeff * __expf(-0.5f * z * z);
    drho = -z / sd * rho;
  }
}

__device__ __forceinline__ void envelope_fast(int kind, float d, float rc, float& f, float& df) {
  if (!(d < rc)) {
    f = 0.f;
    df = 0.f;
    return;
  }
  if (kind == XEQ_CUTOFF_COSINE) {
    float sn, cs;
    sincos_rev(d / (2.f * rc), sn, cs);  // pi d / rc = 2 pi * d / (2 rc), in [0, 1/2) revolutions
    f = 0.5f * (cs + 1.f);
    df = -0.5f * 3.14159265358979f / rc * sn;
  } else {
    envelope<float>(kind, d, rc, f, df);
  }
}

// phase 0: threads [0, 8 KB): lane (k, eh) evaluates rho_k for edges eh and eh + 8 (k fixed per thread);
// threads [192, 208): one lane per edge writes the record (Y_lm, envelope, unit vector, indices).
template <int KS, bool BWD>
__device__ __forceinline__ void phase0(const Msg2Args& a, const float* __restrict__ vec, const float* __restrict__ p0,
                                       const float* __restrict__ p1, int32_t base, int cnt, int64_t n0, int64_t n1,
                                       Smem<KS, BWD>& sm, float wk, float p0k, float p1k) {
  constexpr int KB = 4 * KS;
  static_assert(8 * KB <= 192 || KB == 32, "phase-0 thread map");
  const float rc = (float)a.rs.cutoff;
  const int t = threadIdx.x;
  if (t < 8 * KB) {
    const int k = t % KB, eh = t / KB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = eh + 8 * half;
      float r = 0.f, dr = 0.f;
      if (j < cnt && k < a.rs.num_basis) {
        const int32_t p = base + j;
        const int64_t e = a.perm ? a.perm[p] : p;
        const float vx = vec[3 * e], vy = vec[3 * e + 1], vz = vec[3 * e + 2];
        const float d = sqrtf(vx * vx + vy * vy + vz * vz);
        radial_fast(a.rs.rbf_kind, d, rc, p0k, p1k, wk, r, dr);
      }
      sm.rho[j][k] = r;
      if (BWD) sm.drho[j][k] = dr;
    }
  }
  const int j = t - (KB == 32 ? 240 : 192);  // KB = 32 uses all 256 threads above: the record lanes double up
  if (j >= 0 && j < TE) {
    if (j < cnt) {
      const int32_t p = base + j;
      const int32_t e = a.perm ? a.perm[p] : p;
      EdgeGeom<float> g = edge_geom<float>(vec[3 * (int64_t)e], vec[3 * (int64_t)e + 1], vec[3 * (int64_t)e + 2]);
      float f, df, y1[3], y2[5];
      envelope_fast(a.rs.cutoff_kind, g.d, rc, f, df);
      sph_harm_l12<float>(g, y1, y2);
      sm.y[j][0] = 1.f;
#pragma unroll
      for (int m = 0; m < 3; ++m) sm.y[j][1 + m] = y1[m];
#pragma unroll
      for (int m = 0; m < 5; ++m) sm.y[j][4 + m] = y2[m];
      sm.y[j][9] = f;
      sm.y[j][10] = df;
      sm.g[j][0] = g.x;
      sm.g[j][1] = g.y;
      sm.g[j][2] = g.z;
      sm.g[j][3] = g.d;
      sm.g[j][4] = g.inv_d;
      sm.self[j] = node_of_slot(a.rowptr, n0, n1, p);
      sm.other[j] = (int32_t)a.other_idx[e];
      sm.eid[j] = e;
    } else {
      sm.y[j][9] = 0.f;  // f = 0: the padded column of the filter tile is exactly zero
      sm.y[j][10] = 0.f;
    }
  }
}

// phase 1: filter tile on the matrix cores -> LDS phi[e][c] (and dphi in the reverse pass)
template <int KS, bool BWD>
__device__ __forceinline__ void phase1(const Msg2Args& a, const float (&wa)[TPW][KS], const float* __restrict__ sh_bias,
                                       Smem<KS, BWD>& sm, float* __restrict__ sh_phi, float* __restrict__ sh_dphi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el = lane & 15, g = lane >> 4;
  const float fe = sm.y[el][9], dfe = sm.y[el][10];
  float rb[KS], rdb[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {  // B fragments: f rho_k and (f rho_k)' = f' rho_k + f rho_k'
    const float rho = sm.rho[el][4 * s + g];
    rb[s] = fe * rho;
    if (BWD) rdb[s] = dfe * rho + fe * sm.drho[el][4 * s + g];
  }
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tile = wave + 4 * tt;
    if (tile < a.NT) {
      const int c0 = tile * 16 + 4 * g;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sh_bias + c0);  // zero-padded to NT*16 in LDS
      f32x4 acc = b4 * fe;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tt][s], rb[s], acc, 0, 0, 0);
      *reinterpret_cast<f32x4*>(sh_phi + el * a.HP + c0) = acc;
      if (BWD) {
        f32x4 dacc = b4 * dfe;
#pragma unroll
        for (int s = 0; s < KS; ++s) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tt][s], rdb[s], dacc, 0, 0, 0);
        *reinterpret_cast<f32x4*>(sh_dphi + el * a.HP + c0) = dacc;
      }
    }
  }
}

// A fragments of rbf_lin.weight for this wave's channel tiles: lane (i = l & 15, k = l >> 4)
template <int KS>
__device__ __forceinline__ void load_a_frags(const Msg2Args& a, const float* __restrict__ w_rbf, float (&wa)[TPW][KS]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = a.rs.num_basis;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int c = (wave + 4 * tt) * 16 + (lane & 15);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + (lane >> 4);
      wa[tt][s] = (c < a.H && k < B) ? w_rbf[(int64_t)c * B + k] : 0.f;
    }
  }
}


bool mfma_path_supported(int dtype, int num_basis, int node_dim, const int32_t mul[3]);

}  // namespace xeq
