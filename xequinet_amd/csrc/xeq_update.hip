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
#ifdef XEQ_UV_NO_ST   // development: no U|V stores
    if (row_ok && U[0] == 12345.f) {
#else
    if (row_ok) {
#endif
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
  const int64_t row0 = (int64_t)blockIdx.x * UV_ROWS;
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

  {  // ---- phase A: both norms, 8 lanes per node, the node's row in registers
    const int node = tid >> 3, sub = tid & 7;
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
    UV_LDS_BARRIER();   // the staged parameters (the row loads above stay in flight across it)
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
        *reinterpret_cast<float4*>(a.cat + gn * a.ld_cat + f0) = o;
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
    if (ok && sub == 0) *reinterpret_cast<float4*>(a.stats + 4 * gn) = make_float4(mean, rstd, mean0, r);
  }
  UV_LDS_BARRIER();

  // ---- phase B: the o3.Linear pair on the matrix cores, v and p from the accumulators
#ifdef XEQ_UV_NO_B   // development: norms only
  if (a.n >= 0) return;
#endif
  for (int jj = wave; jj < a.n_jobs; jj += 4) {
    const int l = a.job_l[jj], t = a.job_t[jj];
    const int mul = l == 0 ? m0 : (l == 1 ? m1 : m2);
    if (mul == 128) uv_job<16>(a, xh, XLD, l, t, row0, rows_here, lane);
    else if (mul == 64) uv_job<8>(a, xh, XLD, l, t, row0, rows_here, lane);
    else uv_job<4>(a, xh, XLD, l, t, row0, rows_here, lane);
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
  const dim3 grid((unsigned)((n + UV_ROWS - 1) / UV_ROWS));
  if (mul[0] == 128 && mul[1] == 64 && mul[2] == 32 && node_dim == 128 && do_norm)   // the default model (nn/model.py: 128x0e + 64x1o + 32x2e)
    hipLaunchKernelGGL((k_update_uv_fwd<128, 64, 32, 128, 1>), grid, dim3(256), lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_update_uv_fwd<-1, -1, -1, -1, -1>), grid, dim3(256), lds, (hipStream_t)stream, a);
  XEQ_CHECK_LAUNCH("xeq_update_uv_fwd");
  return XEQ_OK;
}

}  // extern "C"
