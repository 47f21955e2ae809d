// First half of XPainnUpdate.forward (nn/xpainn.py:206-217) as ONE launch on the matrix cores:
//   LayerNorm(s), EquivariantLayerNorm(x)  ->  U = update_U(xhat), V = update_V(xhat)  (o3.Linear, nn/xpainn.py:186-187)
//   ->  v = Invariant(V) (nn/o3layer.py:39-44),  p = EquivariantDot(U, V) (:104-109)
// It replaces xeq_norm_fwd + three library GEMMs + xeq_uv_reduce_fwd: the normalised tile lives in LDS only, U and V are
// reduced to v and p in the accumulator registers that produced them, and what leaves the chip is what the rest of the
// block and the reverse pass read: [shat | v] (the update MLP's input), p, the U|V pair buffer (BT layout, xeq_node.hip),
// and the norm statistics.
//
// Work split: a workgroup (4 waves) owns 32 consecutive nodes.  Phase A: 8 lanes per node compute both norms with the
// node's row in registers and write xhat into LDS as [node][l][m][channel] (the o3.Linear contraction index contiguous).
// Phase B: jobs = (l, tile of 32 output channels), dealt to the waves heaviest first; a job computes U and V of its
// channels for every m with exact-f32 v_mfma_f32_32x32x2_f32 (weights as the A operand: a lane holds four consecutive
// channels of one node per register quad, so every store is 16 bytes), from weights packed in fragment order
// (xeq_mlp_pack of [W_U | W_V] / sqrt(mul), the l = 0 biases as one more k-group).
#include <mutex>
#include <type_traits>

#include "xeq_common.h"

namespace xeq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define UV_SB() __builtin_amdgcn_sched_barrier(0)
#define UV_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int UV_ROWS = 32;
constexpr int UV_MAXX4 = 16;   // D <= 512 (8 lanes x 16 float4 per node row)
constexpr int UV_MAXS4 = 4;    // F <= 128
constexpr int UV_MAXJOBS = 16;

struct UvFwdArgs {
  const float *s, *x, *lnw, *lnb, *eqw, *eqb;
  int do_norm;
  int64_t n;
  int F;
  Irreps ir;
  const float* wp[3];   // packed [W_U | W_V] / sqrt(mul_l): xeq_mlp_pack(n_out = 2 mul_l, k_in = mul_l, transposed = 1)
  int has_bias;         // the l = 0 pack carries [bias_U | bias_V]
  float eps;
  float* cat;
  int64_t ld_cat;
  float* p;
  float* uv;
  float* stats;
  int n_jobs;
  unsigned char job_l[UV_MAXJOBS], job_t[UV_MAXJOBS];
  TileSplit ts;         // workgroup -> (node tile, part of its jobs)
};

__device__ __forceinline__ float sum8(float v) {   // over the 8 lanes of a node
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

// one job: U, V of channels [32 t, 32 t + 32) of block l for every m, then v and p of those channels
template <int G>
__device__ __forceinline__ void uv_job(const UvFwdArgs& a, const float* xh, int XLD, int l, int t, int64_t row0, int rows_here,
                                       int lane) {
  constexpr int GH = G < 8 ? G : 8, NP = G / GH;
  const int i = lane & 31, kh = lane >> 5;
  const int mul = 8 * G, d = 2 * l + 1;
  const int m0 = a.ir.mul[0], m1 = a.ir.mul[1];
  const int base = l == 0 ? 0 : (l == 1 ? m0 : m0 + 3 * m1);   // flat offset of the block in a node row (xhat tile, BT)
  const int goff = l == 0 ? 0 : (l == 1 ? m0 : m0 + m1);       // first gate channel of the block
  const bool row_ok = i < rows_here;
  const float4* wU = reinterpret_cast<const float4*>(a.wp[l]) + (int64_t)t * (G + 1) * 64;
  const float4* wV = reinterpret_cast<const float4*>(a.wp[l]) + (int64_t)(mul / 32 + t) * (G + 1) * 64;
  float* __restrict__ uvb = a.uv + a.n * 2 * base + row0 * d * 2 * mul;   // BT pair buffer, this tile's first row of block l
  // weights: G <= 8 groups stay in registers for every m of the job; G = 16 streams them in quarters of 4 groups, each
  // fetched one quarter ahead into the other half of the register buffer
  float4 wu[GH], wv[GH];
  auto fetch4 = [&](int h, int qq) {   // groups [4 qq, 4 qq + 4) into slots [4 h, 4 h + 4)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wu[4 * h + q] = wU[(4 * qq + q) * 64 + lane];
      wv[4 * h + q] = wV[(4 * qq + q) * 64 + lane];
    }
  };
  if (NP == 1) {
#pragma unroll
    for (int q = 0; q < GH; ++q) {
      wu[q] = wU[q * 64 + lane];
      wv[q] = wV[q * 64 + lane];
    }
  } else {
    fetch4(0, 0);
  }
  const bool bias = l == 0 && a.has_bias;
  const float one_k0 = kh == 0 ? 1.f : 0.f;
  float bu = 0.f, bv = 0.f;
  if (bias) {
    bu = reinterpret_cast<const float*>(wU + G * 64 + lane)[0];
    bv = reinterpret_cast<const float*>(wV + G * 64 + lane)[0];
  }
  f32x16 pacc, vacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    pacc[r] = 0.f;
    vacc[r] = 0.f;
  }
  for (int m = 0; m < d; ++m) {
    f32x16 U, V;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      U[r] = 0.f;
      V[r] = 0.f;
    }
    const float* xs = xh + i * XLD + base + m * mul + 4 * kh;
    auto steps = [&](int slot0, int g0, int ng) {   // ng k-groups from group g0 with the weights in slots slot0..
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q >= ng) break;
        const float4 xv = *reinterpret_cast<const float4*>(xs + 8 * (g0 + q));
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(wu[slot0 + q].x, xv.x, U, 0, 0, 0);
        V = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot0 + q].x, xv.x, V, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(wu[slot0 + q].y, xv.y, U, 0, 0, 0);
        V = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot0 + q].y, xv.y, V, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(wu[slot0 + q].z, xv.z, U, 0, 0, 0);
        V = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot0 + q].z, xv.z, V, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f32_32x32x2f32(wu[slot0 + q].w, xv.w, U, 0, 0, 0);
        V = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot0 + q].w, xv.w, V, 0, 0, 0);
      }
    };
    if constexpr (NP == 1) {
      steps(0, 0, GH);
    } else {   // G = 16: quarters 0..3, the next one (or the next m's first) in flight under each
      fetch4(1, 1);
      UV_SB();
      steps(0, 0, 4);
      UV_SB();
      fetch4(0, 2);
      UV_SB();
      steps(4, 4, 4);
      UV_SB();
      fetch4(1, 3);
      UV_SB();
      steps(0, 8, 4);
      UV_SB();
      if (m + 1 < d) fetch4(0, 0);
      UV_SB();
      steps(4, 12, 4);
      UV_SB();
    }
    if (bias) {
      U = __builtin_amdgcn_mfma_f32_32x32x2f32(bu, one_k0, U, 0, 0, 0);
      V = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, one_k0, V, 0, 0, 0);
    }
    if (row_ok) {
      const unsigned o = (unsigned)(i * d + m) * (unsigned)(2 * mul) + (unsigned)(32 * t + 4 * kh);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(uvb + o + 8u * g) = make_float4(U[4 * g], U[4 * g + 1], U[4 * g + 2], U[4 * g + 3]);
        *reinterpret_cast<float4*>(uvb + o + 8u * g + (unsigned)mul) = make_float4(V[4 * g], V[4 * g + 1], V[4 * g + 2], V[4 * g + 3]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pacc[r] = __builtin_fmaf(U[r], V[r], pacc[r]);
      vacc[r] = __builtin_fmaf(V[r], V[r], vacc[r]);
    }
  }
  if (row_ok) {
    const float e = a.eps, e2 = a.eps * a.eps;
    const int64_t row = row0 + i;
    float* __restrict__ cv = a.cat + row * a.ld_cat + a.F + goff + 32 * t + 4 * kh;
    float* __restrict__ pp = a.p + row * a.ir.C() + goff + 32 * t + 4 * kh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<float4*>(cv + 8 * g) = make_float4(sqrtf(vacc[4 * g] + e2) - e, sqrtf(vacc[4 * g + 1] + e2) - e,
                                                          sqrtf(vacc[4 * g + 2] + e2) - e, sqrtf(vacc[4 * g + 3] + e2) - e);
      *reinterpret_cast<float4*>(pp + 8 * g) = make_float4(pacc[4 * g], pacc[4 * g + 1], pacc[4 * g + 2], pacc[4 * g + 3]);
    }
  }
}

// phase A of the front half: both norms of a tile of ROWS nodes, 8 lanes per node with the node's row in registers; xhat into LDS as
// [node][l][m][channel], shat / the statistics to global memory (part 0 only).  One workgroup barrier inside (the staged parameters);
// the caller places the one behind it.  A node's sums do not depend on ROWS or NT: the 8-lane split of a row is the same.
struct UvNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_loads: called once the node rows are requested, before the barrier (the few-row form requests its job's weights there: a
// wave's loads return in order, and the rows are what phase A waits for)
template <int M0, int M1, int M2, int FF, int NORM, int ROWS, int NT, typename Hook = UvNoHook>
__device__ __forceinline__ void uv_phase_a(const UvFwdArgs& a, float* xh, const float* lnw, const float* lnb, const float* eqw, const float* eqb,
                                           int64_t row0, int rows_here, int part_, int tid, const Hook& after_loads = Hook()) {
  const int m0 = M0 >= 0 ? M0 : a.ir.mul[0], m1 = M1 >= 0 ? M1 : a.ir.mul[1], m2 = M2 >= 0 ? M2 : a.ir.mul[2];
  const int F = FF >= 0 ? FF : a.F;
  const bool do_norm = NORM >= 0 ? (NORM != 0) : (a.do_norm != 0);
  const int D = m0 + 3 * m1 + 5 * m2, C = m0 + m1 + m2, XLD = D + 4;
  {
    const int node = tid >> 3, sub = tid & 7;
    const bool live = 8 * ROWS >= NT || node < ROWS;   // (a workgroup of more than 8 ROWS threads: the rest only meet the barrier)
    const bool ok = node < rows_here;
    const int64_t gn = row0 + min(node, rows_here - 1);
    const float4* sr = reinterpret_cast<const float4*>(a.s + gn * F);
    const float4* xr = reinterpret_cast<const float4*>(a.x + gn * D);
    float4 sv[UV_MAXS4], xv[UV_MAXX4];
#pragma unroll
    for (int k = 0; k < UV_MAXS4; ++k) {
      const int idx = sub + 8 * k;
      const bool v = 4 * idx < F;
      const float4 t = sr[v ? idx : 0];
      sv[k] = make_float4(v ? t.x : 0.f, v ? t.y : 0.f, v ? t.z : 0.f, v ? t.w : 0.f);
    }
#pragma unroll
    for (int k = 0; k < UV_MAXX4; ++k) {
      const int idx = sub + 8 * k;
      const bool v = 4 * idx < D;
      const float4 t = xr[v ? idx : 0];
      xv[k] = make_float4(v ? t.x : 0.f, v ? t.y : 0.f, v ? t.z : 0.f, v ? t.w : 0.f);
    }
    after_loads();
    UV_LDS_BARRIER();   // the staged parameters (the row loads above stay in flight across it)
    if (!live) return;
    float mean = 0.f, rstd = 1.f, mean0 = 0.f, r = 1.f;
    if (do_norm) {
      // nn.LayerNorm over the F scalars (eps 1e-5, biased variance)
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < UV_MAXS4; ++k) acc += (sv[k].x + sv[k].y) + (sv[k].z + sv[k].w);
      mean = sum8(acc) / (float)F;
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < UV_MAXS4; ++k) {
        if (4 * (sub + 8 * k) < F) {
          const float dx = sv[k].x - mean, dy = sv[k].y - mean, dz = sv[k].z - mean, dw = sv[k].w - mean;
          var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
      }
      rstd = 1.f / sqrtf(sum8(var) / (float)F + 1e-5f);
      // EquivariantLayerNorm (nn/o3layer.py:145-171): the 0e channels are centred, one rms over all channels
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < UV_MAXX4; ++k)
        if (4 * (sub + 8 * k) < m0) q += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
      mean0 = m0 > 0 ? sum8(q) / (float)m0 : 0.f;
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < UV_MAXX4; ++k) {
        const float c = 4 * (sub + 8 * k) < m0 ? mean0 : 0.f;   // (past D the row is zero and c is zero)
        const float dx = xv[k].x - c, dy = xv[k].y - c, dz = xv[k].z - c, dw = xv[k].w - c;
        sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      r = 1.f / sqrtf(sum8(sq) / (float)C + 1e-5f);
    }
    // shat -> the scalar columns of [shat | v]
#pragma unroll
    for (int k = 0; k < UV_MAXS4; ++k) {
      const int f0 = 4 * (sub + 8 * k);
      if (ok && f0 < F) {
        float4 o = sv[k];
        if (do_norm) {
          const float4 w = *reinterpret_cast<const float4*>(lnw + f0), b = *reinterpret_cast<const float4*>(lnb + f0);
          o = make_float4((o.x - mean) * rstd * w.x + b.x, (o.y - mean) * rstd * w.y + b.y, (o.z - mean) * rstd * w.z + b.z,
                          (o.w - mean) * rstd * w.w + b.w);
        }
        if (part_ == 0) *reinterpret_cast<float4*>(a.cat + gn * a.ld_cat + f0) = o;
      }
    }
    // xhat -> LDS, e3nn (channel-major, m-minor) re-laid as [l][m][channel]; rows past n are written as zeros
    float* xrow = xh + node * XLD;
#pragma unroll
    for (int k = 0; k < UV_MAXX4; ++k) {
      const int f0 = 4 * (sub + 8 * k);
      if (f0 >= D) continue;
      const float in[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
      if (f0 < m0) {
        float4 o = make_float4(in[0], in[1], in[2], in[3]);
        if (do_norm) {
          const float4 w = *reinterpret_cast<const float4*>(eqw + f0), b = *reinterpret_cast<const float4*>(eqb + f0);
          o = make_float4((o.x - mean0) * r * w.x + b.x, (o.y - mean0) * r * w.y + b.y, (o.z - mean0) * r * w.z + b.z,
                          (o.w - mean0) * r * w.w + b.w);
        }
        *reinterpret_cast<float4*>(xrow + f0) = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const bool is1 = f0 < m0 + 3 * m1;
        const int dl = is1 ? 3 : 5, mul = is1 ? m1 : m2, off = is1 ? m0 : m0 + 3 * m1, u0 = is1 ? m0 : m0 + m1;
        const int rr0 = f0 - off;
        int up = is1 ? rr0 / 3 : rr0 / 5, m = rr0 - up * dl;   // (channel, component) of the first element, then stepped
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float val = in[e];
          if (do_norm) val = val * r * eqw[u0 + up];
          xrow[off + m * mul + up] = ok ? val : 0.f;
          ++m;
          if (m == dl) {
            m = 0;
            ++up;
          }
        }
      }
    }
    if (ok && sub == 0 && part_ == 0) *reinterpret_cast<float4*>(a.stats + 4 * gn) = make_float4(mean, rstd, mean0, r);
    }
}

// M0, M1, M2, FF, NORM >= 0: the layout is a compile-time constant (the default model's instantiation: every region test and
// index division of phase A folds); -1: read from the arguments.
template <int M0, int M1, int M2, int FF, int NORM>
__global__ void __launch_bounds__(256) k_update_uv_fwd(UvFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xh[];   // [32][D + 4]: the normalised tile, [node][l][m][channel]
  const int m0 = M0 >= 0 ? M0 : a.ir.mul[0], m1 = M1 >= 0 ? M1 : a.ir.mul[1], m2 = M2 >= 0 ? M2 : a.ir.mul[2];
  const int F = FF >= 0 ? FF : a.F;
  const bool do_norm = NORM >= 0 ? (NORM != 0) : (a.do_norm != 0);
  const int D = m0 + 3 * m1 + 5 * m2, C = m0 + m1 + m2, XLD = D + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_, part_, parts_;
  a.ts.decode((int)blockIdx.x, tile_, part_, parts_);
  const int64_t row0 = (int64_t)tile_ * UV_ROWS;
  const int rows_here = (int)min((int64_t)UV_ROWS, a.n - row0);
  // norm parameters into LDS first ([ln_w F | ln_b F | eq_w C | eq_b m0], behind the tile): phase A then has ONE round
  // trip to global memory in its dependency chain (the rows), not three
  float* prm = xh + UV_ROWS * XLD;
  float *lnw = prm, *lnb = prm + F, *eqw = prm + 2 * F, *eqb = prm + 2 * F + C;
  if (do_norm) {
    for (int f = tid; f < F; f += 256) {
      lnw[f] = a.lnw[f];
      lnb[f] = a.lnb[f];
    }
    for (int f = tid; f < C; f += 256) eqw[f] = a.eqw[f];
    for (int f = tid; f < m0; f += 256) eqb[f] = a.eqb[f];
  }

  uv_phase_a<M0, M1, M2, FF, NORM, UV_ROWS, 256>(a, xh, lnw, lnb, eqw, eqb, row0, rows_here, part_, tid);
  UV_LDS_BARRIER();

  // ---- phase B: the o3.Linear pair on the matrix cores, v and p from the accumulators
  // split tiles (TileSplit, xeq_common.h): `parts` workgroups share the node tile, each repeats phase A and takes every parts-th job
  const int parts = __builtin_amdgcn_readfirstlane(parts_), part = __builtin_amdgcn_readfirstlane(part_);
  for (int jj = part + parts * wave; jj < a.n_jobs; jj += 4 * parts) {
    const int l = a.job_l[jj], t = a.job_t[jj];
    const int mul = l == 0 ? m0 : (l == 1 ? m1 : m2);
    if (mul == 128) uv_job<16>(a, xh, XLD, l, t, row0, rows_here, lane);
    else if (mul == 64) uv_job<8>(a, xh, XLD, l, t, row0, rows_here, lane);
    else uv_job<4>(a, xh, XLD, l, t, row0, rows_here, lane);
  }
}

// ---- few nodes (MD-sized systems): 16-node tiles, 16 x 16 exact-f32 tiles, two workgroups of 8 waves per tile -----------------------
// v_mfma_f32_16x16x4_f32 fed the k sequence of the 32-row form's instructions rounds to the same bits (a sequential fused-multiply-add
// chain per output element, whatever the tile shape: xeq_linear.hip, scratch/mfma_order), so this form changes no result: a quarter of
// the chain per wave, four times the waves.  Phase A as above (threads 0..127 of both workgroups: within 256 registers, which 16 waves
// in one workgroup are not); phase B: job = (l, 16 output channels), one per wave.
constexpr int UVS_ROWS = 16, UVS_NT = 512, UVS_PARTS = 2;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct UvJobS {   // one job of the few-node forward form: (l, 16 output channels) and its weight fragments, requested whole
  int l, t16, mul;
  float wu[16][2], wv[16][2], bu, bv;
};

__device__ __forceinline__ void uv_job_s_load(const UvFwdArgs& a, int jj, int m0, int m1, int m2, int lane, UvJobS& j) {
  const int j0 = m0 >> 4, j1 = m1 >> 4;
  j.l = jj < j0 ? 0 : (jj < j0 + j1 ? 1 : 2);
  j.t16 = jj - (j.l == 0 ? 0 : (j.l == 1 ? j0 : j0 + j1));
  j.mul = j.l == 0 ? m0 : (j.l == 1 ? m1 : m2);
  const int i = lane & 15, kq = lane >> 4, kh = kq & 1, G = j.mul >> 3;
  const bool sel = (kq >> 1) != 0;
  const int lo = 16 * (j.t16 & 1) + i + 32 * kh;   // this lane's slot in a packed 32-column tile
  const float4* wU = reinterpret_cast<const float4*>(a.wp[j.l]) + (int64_t)(j.t16 >> 1) * (G + 1) * 64 + lo;
  const float4* wV = reinterpret_cast<const float4*>(a.wp[j.l]) + (int64_t)(j.mul / 32 + (j.t16 >> 1)) * (G + 1) * 64 + lo;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int qc = q < G ? q : G - 1;
    const float4 u = wU[qc * 64], v = wV[qc * 64];
    j.wu[q][0] = sel ? u.y : u.x;
    j.wu[q][1] = sel ? u.w : u.z;
    j.wv[q][0] = sel ? v.y : v.x;
    j.wv[q][1] = sel ? v.w : v.z;
  }
  j.bu = j.bv = 0.f;
  if (j.l == 0 && a.has_bias && kq == 0) {
    j.bu = reinterpret_cast<const float*>(wU + G * 64)[0];
    j.bv = reinterpret_cast<const float*>(wV + G * 64)[0];
  }
}

template <int G>
__device__ __forceinline__ void uv_job_s_run(const UvFwdArgs& a, const UvJobS& j, const float* xh, int XLD, int64_t row0, int rows_here,
                                             int lane) {
  const int i = lane & 15, kq = lane >> 4, kh = kq & 1;
  const bool sel = (kq >> 1) != 0;
  const int l = j.l, t16 = j.t16, mul = 8 * G, d = 2 * l + 1;
  const int m0 = a.ir.mul[0], m1 = a.ir.mul[1];
  const int base = l == 0 ? 0 : (l == 1 ? m0 : m0 + 3 * m1);
  const int goff = l == 0 ? 0 : (l == 1 ? m0 : m0 + m1);
  const bool row_ok = i < rows_here;
  const bool bias = l == 0 && a.has_bias;
  const float one_k0 = kq == 0 ? 1.f : 0.f;
  float* __restrict__ uvb = a.uv + a.n * 2 * base + row0 * d * 2 * mul;
  f32x4 pacc = {0.f, 0.f, 0.f, 0.f}, vacc = {0.f, 0.f, 0.f, 0.f};
  for (int m = 0; m < d; ++m) {
    f32x4 U = {0.f, 0.f, 0.f, 0.f}, V = {0.f, 0.f, 0.f, 0.f};
    const float* xs = xh + i * XLD + base + m * mul + 4 * kh;
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + 8 * q);
      const float b0 = sel ? xv.y : xv.x, b1 = sel ? xv.w : xv.z;
      U = __builtin_amdgcn_mfma_f32_16x16x4f32(j.wu[q][0], b0, U, 0, 0, 0);
      V = __builtin_amdgcn_mfma_f32_16x16x4f32(j.wv[q][0], b0, V, 0, 0, 0);
      U = __builtin_amdgcn_mfma_f32_16x16x4f32(j.wu[q][1], b1, U, 0, 0, 0);
      V = __builtin_amdgcn_mfma_f32_16x16x4f32(j.wv[q][1], b1, V, 0, 0, 0);
    }
    if (bias) {
      U = __builtin_amdgcn_mfma_f32_16x16x4f32(j.bu, one_k0, U, 0, 0, 0);
      V = __builtin_amdgcn_mfma_f32_16x16x4f32(j.bv, one_k0, V, 0, 0, 0);
    }
    if (row_ok) {
      const unsigned o = (unsigned)(i * d + m) * (unsigned)(2 * mul) + (unsigned)(16 * t16 + 4 * kq);
      *reinterpret_cast<float4*>(uvb + o) = make_float4(U[0], U[1], U[2], U[3]);
      *reinterpret_cast<float4*>(uvb + o + (unsigned)mul) = make_float4(V[0], V[1], V[2], V[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pacc[r] = __builtin_fmaf(U[r], V[r], pacc[r]);
      vacc[r] = __builtin_fmaf(V[r], V[r], vacc[r]);
    }
  }
  if (row_ok) {
    const float e = a.eps, e2 = a.eps * a.eps;
    const int64_t row = row0 + i;
    *reinterpret_cast<float4*>(a.cat + row * a.ld_cat + a.F + goff + 16 * t16 + 4 * kq) =
        make_float4(sqrtf(vacc[0] + e2) - e, sqrtf(vacc[1] + e2) - e, sqrtf(vacc[2] + e2) - e, sqrtf(vacc[3] + e2) - e);
    *reinterpret_cast<float4*>(a.p + row * a.ir.C() + goff + 16 * t16 + 4 * kq) = make_float4(pacc[0], pacc[1], pacc[2], pacc[3]);
  }
}

template <int M0, int M1, int M2, int FF, int NORM>
__global__ void __launch_bounds__(UVS_NT) k_update_uv_fwd_s(UvFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xh[];   // [16][D + 4], then the norm parameters
  const int m0 = M0 >= 0 ? M0 : a.ir.mul[0], m1 = M1 >= 0 ? M1 : a.ir.mul[1], m2 = M2 >= 0 ? M2 : a.ir.mul[2];
  const int F = FF >= 0 ? FF : a.F;
  const bool do_norm = NORM >= 0 ? (NORM != 0) : (a.do_norm != 0);
  const int D = m0 + 3 * m1 + 5 * m2, C = m0 + m1 + m2, XLD = D + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = (int)blockIdx.x % UVS_PARTS;
  const int64_t row0 = (int64_t)(blockIdx.x / UVS_PARTS) * UVS_ROWS;
  const int rows_here = (int)min((int64_t)UVS_ROWS, a.n - row0);
  // one job per wave of the tile's two workgroups (16 slots for (m0 + m1 + m2) / 16 jobs; more: further rounds); the first job's
  // weights are requested whole, right behind the node rows of phase A: the chain waits for memory once
  const int n_jobs = (m0 + m1 + m2) >> 4;
  int jj = UVS_PARTS * wave + part;
  UvJobS job;
  float* prm = xh + UVS_ROWS * XLD;
  float *lnw = prm, *lnb = prm + F, *eqw = prm + 2 * F, *eqb = prm + 2 * F + C;
  if (do_norm) {
    for (int f = tid; f < F; f += UVS_NT) {
      lnw[f] = a.lnw[f];
      lnb[f] = a.lnb[f];
    }
    for (int f = tid; f < C; f += UVS_NT) eqw[f] = a.eqw[f];
    for (int f = tid; f < m0; f += UVS_NT) eqb[f] = a.eqb[f];
  }
  auto load_first = [&]() {
    if (jj < n_jobs) uv_job_s_load(a, jj, m0, m1, m2, lane, job);
  };
  uv_phase_a<M0, M1, M2, FF, NORM, UVS_ROWS, UVS_NT>(a, xh, lnw, lnb, eqw, eqb, row0, rows_here, part, tid, load_first);
  UV_LDS_BARRIER();
  for (; jj < n_jobs; jj += UVS_PARTS * UVS_NT / 64) {
    if (job.mul == 128) uv_job_s_run<16>(a, job, xh, XLD, row0, rows_here, lane);
    else if (job.mul == 64) uv_job_s_run<8>(a, job, xh, XLD, row0, rows_here, lane);
    else uv_job_s_run<4>(a, job, xh, XLD, row0, rows_here, lane);
    if (jj + UVS_PARTS * UVS_NT / 64 < n_jobs) uv_job_s_load(a, jj + UVS_PARTS * UVS_NT / 64, m0, m1, m2, lane, job);
  }
}

// ================================================================================================================
// Reverse of the front half, one launch:  (dL/dp, dL/dv, dL/dU of the output stage, dL/dshat) -> dL/ds, dL/dx.
//   g_U = g_x_out a_vv + g_p V ;  g_V = g_p U + g_v V / sqrt(sum_m V^2 + eps^2)          (xeq_uv_reduce_bwd)
//   g_xhat = g_U W_U^T + g_V W_V^T  (/ sqrt(mul), per l and m)                            (three library GEMMs)
//   g_s = g_s_out + LN^T(g_shat) ;  g_x = g_x_out + EqLN^T(g_xhat)                        (xeq_norm_bwd)
// Per block l the [g_U | g_V] rows of the 32 nodes are formed in LDS (phase 1), contracted against the packed
// [W_U | W_V]^T on the matrix cores into the g_xhat tile, also in LDS (phase 2); the norms' reverse then runs with 8 lanes
// per node as in the forward kernel (phase 3).  Nothing but g_s and g_x is written.
struct UvBwdArgs {
  const float *uv, *g_p, *g_cat, *g_x_out, *g_s_out, *a, *s, *x, *stats, *lnw, *eqw;
  int64_t ld_cat, ld_a, n;
  int do_norm, F;
  Irreps ir;
  const float* wt[3];   // packed [W_U | W_V]^T / sqrt(mul_l): xeq_mlp_pack(n_out = mul_l, k_in = 2 mul_l, transposed = 0)
  float eps;
  float *g_s, *g_x;
  TileSplit ts;         // workgroup -> (node tile, part of each block's jobs); split form only
  float* g_xhat;        // split form (FUSE = 0): dL/dxhat in BT layout, the norms' reverse is left to xeq_norm_bwd
};

// phase 1 of block L (D_L = 2L+1 components, channels in quads): [g_U | g_V] rows into bf[node][m][2 mul]
template <int D_L, int ROWS = UV_ROWS, int NT = 256>
__device__ __forceinline__ void uvb_form(const UvBwdArgs& a, float* bf, int BLD, int mul, int base, int goff, int F, int Dtot,
                                         int64_t row0, int rows_here, int tid) {
  const int quads = mul >> 2, items = ROWS * quads;
  const float e2 = a.eps * a.eps;
  for (int it = tid; it < items; it += NT) {
    const int node = it / quads, ch = 4 * (it - node * quads);
    const bool ok = node < rows_here;
    const int64_t gn = row0 + min(node, rows_here - 1);
    const float* ur = a.uv + a.n * 2 * base + gn * D_L * 2 * mul + ch;
    float4 U[D_L], V[D_L], gx4[D_L];
#pragma unroll
    for (int m = 0; m < D_L; ++m) {
      U[m] = *reinterpret_cast<const float4*>(ur + m * 2 * mul);
      V[m] = *reinterpret_cast<const float4*>(ur + m * 2 * mul + mul);
      gx4[m] = a.g_x_out ? *reinterpret_cast<const float4*>(a.g_x_out + gn * Dtot + base + ch * D_L + 4 * m)   // 4 D_L contiguous floats
                         : make_float4(0.f, 0.f, 0.f, 0.f);   // no consumer of the block's equivariant output: dL/dx_out = 0
    }
    const float4 gp = *reinterpret_cast<const float4*>(a.g_p + gn * (int64_t)a.ir.C() + goff + ch);
    const float4 gc = *reinterpret_cast<const float4*>(a.g_cat + gn * a.ld_cat + F + goff + ch);
    const float4 av = *reinterpret_cast<const float4*>(a.a + gn * a.ld_a + goff + ch);
    float vv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < D_L; ++m) {
      vv[0] = __builtin_fmaf(V[m].x, V[m].x, vv[0]);
      vv[1] = __builtin_fmaf(V[m].y, V[m].y, vv[1]);
      vv[2] = __builtin_fmaf(V[m].z, V[m].z, vv[2]);
      vv[3] = __builtin_fmaf(V[m].w, V[m].w, vv[3]);
    }
    const float gpv[4] = {gp.x, gp.y, gp.z, gp.w}, avv[4] = {av.x, av.y, av.z, av.w};
    const float gv[4] = {gc.x / sqrtf(vv[0] + e2), gc.y / sqrtf(vv[1] + e2), gc.z / sqrtf(vv[2] + e2), gc.w / sqrtf(vv[3] + e2)};
    float gxf[4 * D_L];   // g_x_out of (channel e, component m) at e D_L + m
#pragma unroll
    for (int m = 0; m < D_L; ++m) {
      gxf[4 * m] = gx4[m].x;
      gxf[4 * m + 1] = gx4[m].y;
      gxf[4 * m + 2] = gx4[m].z;
      gxf[4 * m + 3] = gx4[m].w;
    }
    float* row = bf + node * BLD + ch;
#pragma unroll
    for (int m = 0; m < D_L; ++m) {
      const float u4[4] = {U[m].x, U[m].y, U[m].z, U[m].w}, v4[4] = {V[m].x, V[m].y, V[m].z, V[m].w};
      float gu[4], gw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gu[e] = ok ? __builtin_fmaf(gxf[e * D_L + m], avv[e], gpv[e] * v4[e]) : 0.f;
        gw[e] = ok ? __builtin_fmaf(gpv[e], u4[e], gv[e] * v4[e]) : 0.f;
      }
      *reinterpret_cast<float4*>(row + m * 2 * mul) = make_float4(gu[0], gu[1], gu[2], gu[3]);
      *reinterpret_cast<float4*>(row + m * 2 * mul + mul) = make_float4(gw[0], gw[1], gw[2], gw[3]);
    }
  }
}

// phase 2 of block l: g_xhat[node][l][m][k] = sum_c bf[node][m][c] WT[k][c], c over the 2 mul columns; jobs (m, tile of 32 k)
template <int GQ, bool TO_LDS>   // quarters of 4 k-groups: 2 mul / 32
__device__ __forceinline__ void uvb_contract(const UvBwdArgs& a, const float* bf, int BLD, float* gt, int XLD, int l, int mul,
                                             int base, int wave, int lane, int64_t row0, int rows_here, int part, int parts) {
  const int i = lane & 31, kh = lane >> 5;
  const int T = mul >> 5, d = 2 * l + 1, G = 4 * GQ;
  for (int job = part + parts * wave; job < d * T; job += 4 * parts) {
    const int m = job / T, t = job - m * T;
    const float4* wT = reinterpret_cast<const float4*>(a.wt[l]) + (int64_t)t * (G + 1) * 64;
    const float* bs = bf + i * BLD + m * 2 * mul + 4 * kh;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    float4 W0[4], W1[4];
    auto fetch = [&](float4 (&w)[4], int qq) {
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = wT[(4 * qq + q) * 64 + lane];
    };
    auto steps = [&](const float4 (&w)[4], int qq) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 xv = *reinterpret_cast<const float4*>(bs + 8 * (4 * qq + q));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q].x, xv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q].y, xv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q].z, xv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q].w, xv.w, acc1, 0, 0, 0);
      }
    };
    fetch(W0, 0);
#pragma unroll
    for (int qq = 0; qq < GQ; qq += 2) {
      fetch(W1, qq + 1);
      UV_SB();
      steps(W0, qq);
      UV_SB();
      if (qq + 2 < GQ) fetch(W0, qq + 2);
      UV_SB();
      steps(W1, qq + 1);
      UV_SB();
    }
    float* o = TO_LDS ? gt + i * XLD + base + m * mul + 32 * t + 4 * kh
                      : a.g_xhat + a.n * base + ((row0 + i) * d + m) * mul + 32 * t + 4 * kh;   // BT block l, row (node, m)
    if (TO_LDS || i < rows_here) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(o + 8 * g) = make_float4(acc0[4 * g] + acc1[4 * g], acc0[4 * g + 1] + acc1[4 * g + 1],
                                                            acc0[4 * g + 2] + acc1[4 * g + 2], acc0[4 * g + 3] + acc1[4 * g + 3]);
    }
  }
}

template <int M0, int M1, int M2, int FF, int NORM, bool FUSE>
__global__ void __launch_bounds__(256) k_update_uv_bwd(UvBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int m0 = M0 >= 0 ? M0 : a.ir.mul[0], m1 = M1 >= 0 ? M1 : a.ir.mul[1], m2 = M2 >= 0 ? M2 : a.ir.mul[2];
  const int F = FF >= 0 ? FF : a.F;
  const bool do_norm = NORM >= 0 ? (NORM != 0) : (a.do_norm != 0);
  const int D = m0 + 3 * m1 + 5 * m2, C = m0 + m1 + m2, XLD = D + 4;
  const int bw = max(2 * m0, max(6 * m1, 10 * m2)), BLD = bw + 4;   // widest [g_U | g_V] row of a block
  float* gt = lds;                                    // [32][XLD]  g_xhat tile, [node][l][m][k] (fused form only)
  float* bf = gt + (FUSE ? UV_ROWS * XLD : 0);        // [32][BLD]  [g_U | g_V] rows of the current block
  float* lnw = bf + UV_ROWS * BLD;        // [F], then eq_w [C]
  float* eqw = lnw + F;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_, part_, parts_;
  a.ts.decode((int)blockIdx.x, tile_, part_, parts_);
  const int part = __builtin_amdgcn_readfirstlane(part_), parts = __builtin_amdgcn_readfirstlane(parts_);
  const int64_t row0 = (int64_t)tile_ * UV_ROWS;
  const int rows_here = (int)min((int64_t)UV_ROWS, a.n - row0);
  if (FUSE && do_norm) {
    for (int f = tid; f < F; f += 256) lnw[f] = a.lnw[f];
    for (int f = tid; f < C; f += 256) eqw[f] = a.eqw[f];
  }
  // ---- phases 1 and 2, block by block
  if (m0 > 0) {
    uvb_form<1>(a, bf, BLD, m0, 0, 0, F, D, row0, rows_here, tid);
    UV_LDS_BARRIER();
    if (m0 == 128) uvb_contract<8, FUSE>(a, bf, BLD, gt, XLD, 0, m0, 0, wave, lane, row0, rows_here, part, parts);
    else if (m0 == 64) uvb_contract<4, FUSE>(a, bf, BLD, gt, XLD, 0, m0, 0, wave, lane, row0, rows_here, part, parts);
    else uvb_contract<2, FUSE>(a, bf, BLD, gt, XLD, 0, m0, 0, wave, lane, row0, rows_here, part, parts);
    UV_LDS_BARRIER();
  }
  if (m1 > 0) {
    uvb_form<3>(a, bf, BLD, m1, m0, m0, F, D, row0, rows_here, tid);
    UV_LDS_BARRIER();
    if (m1 == 128) uvb_contract<8, FUSE>(a, bf, BLD, gt, XLD, 1, m1, m0, wave, lane, row0, rows_here, part, parts);
    else if (m1 == 64) uvb_contract<4, FUSE>(a, bf, BLD, gt, XLD, 1, m1, m0, wave, lane, row0, rows_here, part, parts);
    else uvb_contract<2, FUSE>(a, bf, BLD, gt, XLD, 1, m1, m0, wave, lane, row0, rows_here, part, parts);
    UV_LDS_BARRIER();
  }
  if (m2 > 0) {
    uvb_form<5>(a, bf, BLD, m2, m0 + 3 * m1, m0 + m1, F, D, row0, rows_here, tid);
    UV_LDS_BARRIER();
    if (m2 == 128) uvb_contract<8, FUSE>(a, bf, BLD, gt, XLD, 2, m2, m0 + 3 * m1, wave, lane, row0, rows_here, part, parts);
    else if (m2 == 64) uvb_contract<4, FUSE>(a, bf, BLD, gt, XLD, 2, m2, m0 + 3 * m1, wave, lane, row0, rows_here, part, parts);
    else uvb_contract<2, FUSE>(a, bf, BLD, gt, XLD, 2, m2, m0 + 3 * m1, wave, lane, row0, rows_here, part, parts);
    UV_LDS_BARRIER();
  }
  if (!FUSE) return;
  // ---- phase 3: reverse of both norms, 8 lanes per node
  const int node = tid >> 3, sub = tid & 7;
  if (node >= rows_here) return;
  const int64_t gn = row0 + node;
  const float4* sr = reinterpret_cast<const float4*>(a.s + gn * F);
  const float4* xr = reinterpret_cast<const float4*>(a.x + gn * D);
  const float4* gsr = reinterpret_cast<const float4*>(a.g_cat + gn * a.ld_cat);
  const float4* rsr = reinterpret_cast<const float4*>(a.g_s_out + gn * F);
  const float4* rxr = a.g_x_out ? reinterpret_cast<const float4*>(a.g_x_out + gn * D) : nullptr;
  const float4 st = *reinterpret_cast<const float4*>(a.stats + 4 * gn);
  const float mean = st.x, rstd = st.y, mean0 = st.z, r = st.w;
  // g_s
  {
    float4 dy[UV_MAXS4], yh[UV_MAXS4];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < UV_MAXS4; ++k) {
      const int idx = sub + 8 * k, f0 = 4 * idx;
      const bool v = f0 < F;
      const float4 g = gsr[v ? idx : 0], sv = sr[v ? idx : 0];
      if (do_norm) {
        const float4 w = *reinterpret_cast<const float4*>(lnw + (v ? f0 : 0));
        dy[k] = make_float4(v ? g.x * w.x : 0.f, v ? g.y * w.y : 0.f, v ? g.z * w.z : 0.f, v ? g.w * w.w : 0.f);
        yh[k] = make_float4(v ? (sv.x - mean) * rstd : 0.f, v ? (sv.y - mean) * rstd : 0.f, v ? (sv.z - mean) * rstd : 0.f,
                            v ? (sv.w - mean) * rstd : 0.f);
        a1 += (dy[k].x + dy[k].y) + (dy[k].z + dy[k].w);
        a2 += (dy[k].x * yh[k].x + dy[k].y * yh[k].y) + (dy[k].z * yh[k].z + dy[k].w * yh[k].w);
      } else {
        dy[k] = g;
      }
    }
    if (do_norm) {
      a1 = sum8(a1) / (float)F;
      a2 = sum8(a2) / (float)F;
    }
#pragma unroll
    for (int k = 0; k < UV_MAXS4; ++k) {
      const int idx = sub + 8 * k;
      if (4 * idx >= F) continue;
      const float4 rs = rsr[idx];
      float4 o = dy[k];
      if (do_norm)
        o = make_float4(rstd * (o.x - a1 - yh[k].x * a2), rstd * (o.y - a1 - yh[k].y * a2), rstd * (o.z - a1 - yh[k].z * a2),
                        rstd * (o.w - a1 - yh[k].w * a2));
      *reinterpret_cast<float4*>(a.g_s + gn * F + 4 * idx) = make_float4(o.x + rs.x, o.y + rs.y, o.z + rs.z, o.w + rs.w);
    }
  }
  // g_x: gather this node's g_xhat row from the tile in e3nn order, times the affine weight
  {
    const float* grow = gt + node * XLD;
    float4 gw[UV_MAXX4], xc[UV_MAXX4];   // g_xhat eq_w and the centred row
    float dotp = 0.f, gsum = 0.f;
#pragma unroll
    for (int k = 0; k < UV_MAXX4; ++k) {
      const int idx = sub + 8 * k, f0 = 4 * idx;
      if (f0 >= D) {
        gw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        xc[k] = gw[k];
        continue;
      }
      const float4 xv = xr[idx];
      float g[4];
      if (f0 < m0) {
        const float4 t = *reinterpret_cast<const float4*>(grow + f0);
        g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
        if (do_norm) {
          const float4 w = *reinterpret_cast<const float4*>(eqw + f0);
          g[0] *= w.x; g[1] *= w.y; g[2] *= w.z; g[3] *= w.w;
        }
        xc[k] = make_float4(xv.x - mean0, xv.y - mean0, xv.z - mean0, xv.w - mean0);
      } else {
        const bool is1 = f0 < m0 + 3 * m1;
        const int dl = is1 ? 3 : 5, mul = is1 ? m1 : m2, off = is1 ? m0 : m0 + 3 * m1, u0 = is1 ? m0 : m0 + m1;
        const int rr0 = f0 - off;
        int up = is1 ? rr0 / 3 : rr0 / 5, m = rr0 - up * dl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e] = grow[off + m * mul + up];
          if (do_norm) g[e] *= eqw[u0 + up];
          ++m;
          if (m == dl) {
            m = 0;
            ++up;
          }
        }
        xc[k] = xv;
      }
      gw[k] = make_float4(g[0], g[1], g[2], g[3]);
      dotp += (gw[k].x * xc[k].x + gw[k].y * xc[k].y) + (gw[k].z * xc[k].z + gw[k].w * xc[k].w);
    }
    float coef = 0.f, gmean = 0.f;
    if (do_norm) {
      coef = sum8(dotp) * r * r * r / (float)C;
#pragma unroll
      for (int k = 0; k < UV_MAXX4; ++k)
        if (4 * (sub + 8 * k) < m0)
          gsum += ((r * gw[k].x - coef * xc[k].x) + (r * gw[k].y - coef * xc[k].y)) + ((r * gw[k].z - coef * xc[k].z) + (r * gw[k].w - coef * xc[k].w));
      gmean = m0 > 0 ? sum8(gsum) / (float)m0 : 0.f;
    }
#pragma unroll
    for (int k = 0; k < UV_MAXX4; ++k) {
      const int idx = sub + 8 * k, f0 = 4 * idx;
      if (f0 >= D) continue;
      const float4 rx = rxr ? rxr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 o = gw[k];
      if (do_norm) {
        const float gm = f0 < m0 ? gmean : 0.f;
        o = make_float4(r * o.x - coef * xc[k].x - gm, r * o.y - coef * xc[k].y - gm, r * o.z - coef * xc[k].z - gm,
                        r * o.w - coef * xc[k].w - gm);
      }
      *reinterpret_cast<float4*>(a.g_x + gn * D + f0) = make_float4(o.x + rx.x, o.y + rx.y, o.z + rx.z, o.w + rx.w);
    }
  }
}

// ---- few nodes: the split reverse form (dL/dxhat out, the norms' reverse left to xeq_norm_bwd) on 16-node tiles ------------------------
// Bit-equal to the 32-row form (see k_update_uv_fwd_s).  The [g_U | g_V] rows of ALL three blocks are formed at once (one round trip to
// global memory, one barrier: 16 rows x 960 floats of LDS for the default layout), then every (l, m, 16 k) product is a job: 30 for the
// default layout, dealt heaviest first to the 16 waves of the tile's two workgroups, the weight fragments of a job requested whole.
struct UvbJobS {   // one job of the few-node reverse form: (l, m, 16 k) and its weight fragments (even | odd accumulator), requested whole
  int l, m, t16, mul;
  float2 w[32];
};

__device__ __forceinline__ void uvb_job_s_load(const UvBwdArgs& a, int jj, int m0, int m1, int m2, int lane, UvbJobS& j) {
  const int j0 = m0 >> 4, j1 = 3 * (m1 >> 4);
  j.l = jj < j0 ? 0 : (jj < j0 + j1 ? 1 : 2);
  const int r = jj - (j.l == 0 ? 0 : (j.l == 1 ? j0 : j0 + j1));
  j.mul = j.l == 0 ? m0 : (j.l == 1 ? m1 : m2);
  const int T = j.mul >> 4;
  j.m = r / T;
  j.t16 = r - j.m * T;
  const int i = lane & 15, kq = lane >> 4, kh = kq & 1, sel = kq >> 1, G = j.mul >> 2;   // k = 2 mul: G groups of 8
  const float2* wT = reinterpret_cast<const float2*>(reinterpret_cast<const float4*>(a.wt[j.l]) + (int64_t)(j.t16 >> 1) * (G + 1) * 64 +
                                                     16 * (j.t16 & 1) + i + 32 * kh) + sel;
#pragma unroll
  for (int q = 0; q < 32; ++q) j.w[q] = wT[(q < G ? q : G - 1) * 128];
}

template <int G>
__device__ __forceinline__ void uvb_job_s_run(const UvBwdArgs& a, const UvbJobS& j, const float* bf, int BLD, int base, int lane,
                                              int64_t row0, int rows_here) {
  const int i = lane & 15, kq = lane >> 4, kh = kq & 1, sel = kq >> 1;
  const int d = 2 * j.l + 1, mul = 4 * G;   // G groups of 8 over k = 2 mul
  const float* bs = bf + i * BLD + j.m * 2 * mul + 4 * kh + 2 * sel;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const float2 xv = *reinterpret_cast<const float2*>(bs + 8 * q);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(j.w[q].x, xv.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(j.w[q].y, xv.y, acc1, 0, 0, 0);
  }
  if (i < rows_here)
    *reinterpret_cast<float4*>(a.g_xhat + a.n * base + ((row0 + i) * d + j.m) * mul + 16 * j.t16 + 4 * kq) =
        make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
}

template <int M0, int M1, int M2, int FF>
__global__ void __launch_bounds__(UVS_NT) k_update_uv_bwd_s(UvBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int m0 = M0 >= 0 ? M0 : a.ir.mul[0], m1 = M1 >= 0 ? M1 : a.ir.mul[1], m2 = M2 >= 0 ? M2 : a.ir.mul[2];
  const int F = FF >= 0 ? FF : a.F;
  const int D = m0 + 3 * m1 + 5 * m2;
  const int BLD0 = 2 * m0 + 4, BLD1 = 6 * m1 + 4, BLD2 = 10 * m2 + 4;
  float* bf0 = lds;
  float* bf1 = bf0 + UVS_ROWS * BLD0;
  float* bf2 = bf1 + UVS_ROWS * BLD1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = (int)blockIdx.x % UVS_PARTS;
  const int64_t row0 = (int64_t)(blockIdx.x / UVS_PARTS) * UVS_ROWS;
  const int rows_here = (int)min((int64_t)UVS_ROWS, a.n - row0);
  // jobs (l, m, t16), heaviest block first, one per wave of the tile's two workgroups and round; the weights of a wave's first two
  // jobs are requested before the rows are formed
  constexpr int STEP = UVS_PARTS * UVS_NT / 64;
  const int n_jobs = (m0 >> 4) + 3 * (m1 >> 4) + 5 * (m2 >> 4);
  int jj = UVS_PARTS * wave + part;
  UvbJobS ja, jb;
  if (jj < n_jobs) uvb_job_s_load(a, jj, m0, m1, m2, lane, ja);
  constexpr bool AHEAD2 = M0 >= 0;   // (the layout-generic instantiation has no registers for a second set)
  if (AHEAD2 && jj + STEP < n_jobs) uvb_job_s_load(a, jj + STEP, m0, m1, m2, lane, jb);
  if (m0 > 0) uvb_form<1, UVS_ROWS, UVS_NT>(a, bf0, BLD0, m0, 0, 0, F, D, row0, rows_here, tid);
  if (m1 > 0) uvb_form<3, UVS_ROWS, UVS_NT>(a, bf1, BLD1, m1, m0, m0, F, D, row0, rows_here, tid);
  if (m2 > 0) uvb_form<5, UVS_ROWS, UVS_NT>(a, bf2, BLD2, m2, m0 + 3 * m1, m0 + m1, F, D, row0, rows_here, tid);
  UV_LDS_BARRIER();
  auto run = [&](const UvbJobS& j) {
    const float* bf = j.l == 0 ? bf0 : (j.l == 1 ? bf1 : bf2);
    const int BLD = j.l == 0 ? BLD0 : (j.l == 1 ? BLD1 : BLD2);
    const int base = j.l == 0 ? 0 : (j.l == 1 ? m0 : m0 + 3 * m1);
    if (j.mul == 128) uvb_job_s_run<32>(a, j, bf, BLD, base, lane, row0, rows_here);
    else if (j.mul == 64) uvb_job_s_run<16>(a, j, bf, BLD, base, lane, row0, rows_here);
    else uvb_job_s_run<8>(a, j, bf, BLD, base, lane, row0, rows_here);
  };
  for (; jj < n_jobs; jj += 2 * STEP) {
    run(ja);
    if (jj + 2 * STEP < n_jobs) uvb_job_s_load(a, jj + 2 * STEP, m0, m1, m2, lane, ja);
    if (jj + STEP < n_jobs) {
      if (!AHEAD2) uvb_job_s_load(a, jj + STEP, m0, m1, m2, lane, jb);
      run(jb);
      if (AHEAD2 && jj + 3 * STEP < n_jobs) uvb_job_s_load(a, jj + 3 * STEP, m0, m1, m2, lane, jb);
    }
  }
}

static bool uv_shape_ok(int node_dim, const Irreps& ir) {
  for (int l = 0; l < 3; ++l)
    if (!(ir.mul[l] == 0 || ir.mul[l] == 32 || ir.mul[l] == 64 || ir.mul[l] == 128)) return false;
  int jobs = (ir.mul[0] + ir.mul[1] + ir.mul[2]) / 32;
  return ir.C() > 0 && jobs <= UV_MAXJOBS && node_dim > 0 && node_dim % 4 == 0 && node_dim <= 32 * UV_MAXS4 && ir.D() <= 480;   // tile + parameters within 64 KB of LDS
}

}  // namespace xeq

using namespace xeq;

extern "C" {

int xeq_update_uv_supported(int dtype, int node_dim, const int32_t mul[3]) {
  Irreps ir{{mul[0], mul[1], mul[2]}};
  return dtype == XEQ_F32 && mul[0] >= 0 && mul[1] >= 0 && mul[2] >= 0 && uv_shape_ok(node_dim, ir);
}

int xeq_update_uv_fwd(const float* s, const float* x, const float* ln_w, const float* ln_b, const float* eq_w, const float* eq_b,
                      int64_t n, int node_dim, const int32_t mul[3], int do_norm, const float* w_packed0, const float* w_packed1,
                      const float* w_packed2, int has_bias, double eps, float* cat, int64_t ld_cat, float* p, float* uv_bt,
                      float* stats, void* stream) {
  XEQ_CHECK_ARG(xeq_update_uv_supported(XEQ_F32, node_dim, mul), "xeq_update_uv_fwd: unsupported layout (node_dim %d, mul %d %d %d)",
                node_dim, mul[0], mul[1], mul[2]);
  Irreps ir{{mul[0], mul[1], mul[2]}};
  XEQ_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) * UV_ROWS / 64, "xeq_update_uv_fwd: n = %lld out of range", (long long)n);
  XEQ_CHECK_ARG(ld_cat >= node_dim + ir.C() && ld_cat % 4 == 0, "xeq_update_uv_fwd: ld_cat = %lld does not hold [shat | v] in 16-byte rows",
                (long long)ld_cat);
  if (n == 0) return XEQ_OK;
  const float* wp[3] = {w_packed0, w_packed1, w_packed2};
  for (int l = 0; l < 3; ++l) XEQ_CHECK_ARG(mul[l] == 0 || wp[l], "xeq_update_uv_fwd: packed weights of l = %d missing", l);
  XEQ_CHECK_ARG(s && x && cat && p && uv_bt && stats && (!do_norm || (ln_w && ln_b && eq_w && eq_b)), "xeq_update_uv_fwd: null buffer");
  UvFwdArgs a;
  a.s = s; a.x = x; a.lnw = ln_w; a.lnb = ln_b; a.eqw = eq_w; a.eqb = eq_b;
  a.do_norm = do_norm; a.n = n; a.F = node_dim; a.ir = ir;
  for (int l = 0; l < 3; ++l) a.wp[l] = wp[l];
  a.has_bias = has_bias; a.eps = (float)eps;
  a.cat = cat; a.ld_cat = ld_cat; a.p = p; a.uv = uv_bt; a.stats = stats;
  // jobs, heaviest first (cost ~ (2l+1) mul_l MFMA steps), dealt round-robin to the 4 waves
  int order[3] = {0, 1, 2};
  auto cost = [&](int l) { return (2 * l + 1) * mul[l]; };
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (cost(order[j]) > cost(order[i])) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  a.n_jobs = 0;
  for (int k = 0; k < 3; ++k)
    for (int t = 0; t < mul[order[k]] / 32; ++t) {
      a.job_l[a.n_jobs] = (unsigned char)order[k];
      a.job_t[a.n_jobs] = (unsigned char)t;
      ++a.n_jobs;
    }
  const size_t lds = ((size_t)UV_ROWS * (ir.D() + 4) + 2 * node_dim + ir.C() + mul[0]) * sizeof(float);
  const int64_t tiles = (n + UV_ROWS - 1) / UV_ROWS;
  a.ts = tile_split(tiles, a.n_jobs);
  const dim3 grid(a.ts.grid(tiles));
  const bool dflt = mul[0] == 128 && mul[1] == 64 && mul[2] == 32 && node_dim == 128 && do_norm;   // the default model (nn/model.py: 128x0e + 64x1o + 32x2e)
  if (n <= xeq_small_rows()) {   // few nodes: 16-node tiles, 16 waves each (bit-equal results)
    const size_t lds_s = ((size_t)UVS_ROWS * (ir.D() + 4) + 2 * node_dim + ir.C() + mul[0]) * sizeof(float);
    const dim3 grid_s((unsigned)((n + UVS_ROWS - 1) / UVS_ROWS) * UVS_PARTS);
    if (dflt) hipLaunchKernelGGL((k_update_uv_fwd_s<128, 64, 32, 128, 1>), grid_s, dim3(UVS_NT), lds_s, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_update_uv_fwd_s<-1, -1, -1, -1, -1>), grid_s, dim3(UVS_NT), lds_s, (hipStream_t)stream, a);
  } else if (dflt)
    hipLaunchKernelGGL((k_update_uv_fwd<128, 64, 32, 128, 1>), grid, dim3(256), lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_update_uv_fwd<-1, -1, -1, -1, -1>), grid, dim3(256), lds, (hipStream_t)stream, a);
  XEQ_CHECK_LAUNCH("xeq_update_uv_fwd");
  return XEQ_OK;
}

int xeq_update_uv_bwd(const float* uv_bt, const float* g_p, const float* g_cat, int64_t ld_cat, const float* g_x_out,
                      const float* g_s_out, const float* a, int64_t ld_a, const float* s, const float* x, const float* stats,
                      const float* ln_w, const float* eq_w, int64_t n, int node_dim, const int32_t mul[3], int do_norm,
                      const float* wt_packed0, const float* wt_packed1, const float* wt_packed2, double eps, float* g_s, float* g_x,
                      float* g_xhat_bt, void* stream) {
  XEQ_CHECK_ARG(xeq_update_uv_supported(XEQ_F32, node_dim, mul), "xeq_update_uv_bwd: unsupported layout (node_dim %d, mul %d %d %d)",
                node_dim, mul[0], mul[1], mul[2]);
  Irreps ir{{mul[0], mul[1], mul[2]}};
  XEQ_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) * UV_ROWS / 64, "xeq_update_uv_bwd: n = %lld out of range", (long long)n);
  XEQ_CHECK_ARG(ld_cat >= node_dim + ir.C() && ld_cat % 4 == 0 && ld_a >= ir.C() && ld_a % 4 == 0,
                "xeq_update_uv_bwd: row strides (%lld, %lld) do not hold the rows in 16-byte units", (long long)ld_cat, (long long)ld_a);
  if (n == 0) return XEQ_OK;
  const float* wt[3] = {wt_packed0, wt_packed1, wt_packed2};
  for (int l = 0; l < 3; ++l) XEQ_CHECK_ARG(mul[l] == 0 || wt[l], "xeq_update_uv_bwd: packed weights of l = %d missing", l);
  const bool fuse = g_xhat_bt == nullptr;
  XEQ_CHECK_ARG(uv_bt && g_p && g_cat && a, "xeq_update_uv_bwd: null buffer");   // g_x_out may be NULL: zero
  XEQ_CHECK_ARG(!fuse || (g_s_out && s && x && stats && g_s && g_x && (!do_norm || (ln_w && eq_w))), "xeq_update_uv_bwd: null buffer (fused form)");
  UvBwdArgs b;
  b.uv = uv_bt; b.g_p = g_p; b.g_cat = g_cat; b.g_x_out = g_x_out; b.g_s_out = g_s_out; b.a = a; b.s = s; b.x = x; b.stats = stats;
  b.lnw = ln_w; b.eqw = eq_w; b.ld_cat = ld_cat; b.ld_a = ld_a; b.n = n; b.do_norm = do_norm; b.F = node_dim; b.ir = ir;
  for (int l = 0; l < 3; ++l) b.wt[l] = wt[l];
  b.eps = (float)eps; b.g_s = g_s; b.g_x = g_x; b.g_xhat = g_xhat_bt;
  int bw = 2 * mul[0];
  if (6 * mul[1] > bw) bw = 6 * mul[1];
  if (10 * mul[2] > bw) bw = 10 * mul[2];
  const size_t lds = ((fuse ? (size_t)UV_ROWS * (ir.D() + 4) : 0) + (size_t)UV_ROWS * (bw + 4) + node_dim + ir.C()) * sizeof(float);
  int min_jobs = 1 << 30;   // jobs of the smallest block: (2l+1) mul_l / 32
  for (int l = 0; l < 3; ++l)
    if (mul[l] > 0 && (2 * l + 1) * mul[l] / 32 < min_jobs) min_jobs = (2 * l + 1) * mul[l] / 32;
  const int64_t tiles = (n + UV_ROWS - 1) / UV_ROWS;
  b.ts = tile_split(tiles, fuse ? 1 : min_jobs);
  const dim3 grid(b.ts.grid(tiles));
  static std::once_flag attr_once;   // more than 64 KB of dynamic LDS (fused form): opt in once per process
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_uv_bwd<128, 64, 32, 128, 1, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_uv_bwd<-1, -1, -1, -1, -1, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_err = e0 != hipSuccess ? e0 : e1;
  });
  XEQ_CHECK_ARG(attr_err == hipSuccess, "xeq_update_uv_bwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_err));
  const bool dflt = mul[0] == 128 && mul[1] == 64 && mul[2] == 32 && node_dim == 128 && do_norm;
  const size_t lds_s = (size_t)UVS_ROWS * (2 * mul[0] + 6 * mul[1] + 10 * mul[2] + 12) * sizeof(float);
  if (!fuse && n <= xeq_small_rows() && lds_s <= 64 * 1024) {   // few nodes: 16-node tiles, 16 waves each (bit-equal results)
    const dim3 grid_s((unsigned)((n + UVS_ROWS - 1) / UVS_ROWS) * UVS_PARTS);
    if (mul[0] == 128 && mul[1] == 64 && mul[2] == 32 && node_dim == 128)
      hipLaunchKernelGGL((k_update_uv_bwd_s<128, 64, 32, 128>), grid_s, dim3(UVS_NT), lds_s, (hipStream_t)stream, b);
    else
      hipLaunchKernelGGL((k_update_uv_bwd_s<-1, -1, -1, -1>), grid_s, dim3(UVS_NT), lds_s, (hipStream_t)stream, b);
  } else if (fuse) {
    if (dflt) hipLaunchKernelGGL((k_update_uv_bwd<128, 64, 32, 128, 1, true>), grid, dim3(256), lds, (hipStream_t)stream, b);
    else hipLaunchKernelGGL((k_update_uv_bwd<-1, -1, -1, -1, -1, true>), grid, dim3(256), lds, (hipStream_t)stream, b);
  } else {
    if (dflt) hipLaunchKernelGGL((k_update_uv_bwd<128, 64, 32, 128, 1, false>), grid, dim3(256), lds, (hipStream_t)stream, b);
    else hipLaunchKernelGGL((k_update_uv_bwd<-1, -1, -1, -1, -1, false>), grid, dim3(256), lds, (hipStream_t)stream, b);
  }
  XEQ_CHECK_LAUNCH("xeq_update_uv_bwd");
  return XEQ_OK;
}

}  // extern "C"
