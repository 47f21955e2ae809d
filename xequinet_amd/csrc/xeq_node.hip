// Node-side fused elementwise kernels around the dense contractions of
// XPainnMessage / XPainnUpdate (nn/xpainn.py:128-139, 206-231).
//
// Internal "BT" layout of equivariant intermediates (xhat, U|V and their gradients):
// block-major over l, then node, then m, then channel:
//   addr(n, u' in block l, m) = N * base_l + (n * d_l + m) * W_l + u',  d_l = 2l+1,
// with W_l = mul_l (xhat) or 2 mul_l (the U|V pair buffer, U in columns [0,mul), V in
// [mul, 2 mul)).  Every block is a plain row-major [N d_l, W_l] matrix, so o3.Linear is
// three ordinary GEMMs without transposes, and channel-on-lane accesses are coalesced.
// Tensors that cross the module boundary (s, x and their gradients) keep the
// reference's e3nn mul_ir layout.
#include <stdlib.h>

#include "xeq_common.h"

namespace xeq {

struct BT {
  Irreps ir;
  int64_t N;
  // element offset of (n, u, m); `pair` = 1 for the U|V buffer (row width 2 mul), col0 = 0 (U) or mul (V)
  __device__ __forceinline__ int64_t at(int64_t n, int u, int m, int pair, int vcol) const {
    int l, up;
    if (u < ir.mul[0]) { l = 0; up = u; }
    else if (u < ir.mul[0] + ir.mul[1]) { l = 1; up = u - ir.mul[0]; }
    else { l = 2; up = u - ir.mul[0] - ir.mul[1]; }
    const int w = (pair + 1) * ir.mul[l];
    const int64_t base = (l == 0 ? 0 : (l == 1 ? (int64_t)ir.mul[0] : (int64_t)ir.mul[0] + 3 * ir.mul[1])) * (pair + 1);
    return N * base + (n * (2 * l + 1) + m) * w + (vcol ? ir.mul[l] : 0) + up;
  }
};

__device__ __forceinline__ void chan_of_flat(const Irreps& ir, int f, int& u, int& m) {
  if (f < ir.mul[0]) { u = f; m = 0; }
  else if (f < ir.mul[0] + 3 * ir.mul[1]) { int r = f - ir.mul[0]; u = ir.mul[0] + r / 3; m = r % 3; }
  else { int r = f - ir.mul[0] - 3 * ir.mul[1]; u = ir.mul[0] + ir.mul[1] + r / 5; m = r % 5; }
}

// ---- LayerNorm(s) + EquivariantLayerNorm(x): one wave per node ------------------------------
// shat row stride ld_s (so it can land in the [shat | v] buffer of the update MLP); xhat in BT.
// stats[n] = (ln_mean, ln_rstd, eq_mean0, eq_r).  do_norm = 0: identity (layer_norm=False).
template <typename T>
__global__ void k_norm_fwd(const T* __restrict__ s, const T* __restrict__ x, const T* __restrict__ lnw,
                           const T* __restrict__ lnb, const T* __restrict__ eqw, const T* __restrict__ eqb, int64_t N,
                           int F, Irreps ir, int do_norm, T* __restrict__ shat, int64_t ld_s, T* __restrict__ xhat_bt,
                           T* __restrict__ stats) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (n >= N) return;
  BT bt{ir, N};
  const T* sr = s + n * F;
  const T* xr = x + n * D;
  if (!do_norm) {
    for (int f = lane; f < F; f += 64) shat[n * ld_s + f] = sr[f];
    for (int f = lane; f < D; f += 64) {
      int u, m;
      chan_of_flat(ir, f, u, m);
      xhat_bt[bt.at(n, u, m, 0, 0)] = xr[f];
    }
    return;
  }
  // LayerNorm over the F scalars (nn.LayerNorm, eps 1e-5, biased variance)
  T a = T(0);
  for (int f = lane; f < F; f += 64) a += sr[f];
  const T mean = wave_sum<T>(a) / T(F);
  T v = T(0);
  for (int f = lane; f < F; f += 64) { T d = sr[f] - mean; v += d * d; }
  const T rstd = T(1) / sqrt_<T>(wave_sum<T>(v) / T(F) + T(1e-5));
  for (int f = lane; f < F; f += 64) shat[n * ld_s + f] = (sr[f] - mean) * rstd * lnw[f] + lnb[f];
  // EquivariantLayerNorm (nn/o3layer.py:145-171)
  T q = T(0);
  for (int f = lane; f < m0; f += 64) q += xr[f];
  const T mean0 = m0 > 0 ? wave_sum<T>(q) / T(m0) : T(0);
  T sq = T(0);
  for (int f = lane; f < D; f += 64) { T d = xr[f] - (f < m0 ? mean0 : T(0)); sq += d * d; }
  const T r = T(1) / sqrt_<T>(wave_sum<T>(sq) / T(C) + T(1e-5));
  for (int f = lane; f < D; f += 64) {
    int u, m;
    chan_of_flat(ir, f, u, m);
    T val = (xr[f] - (f < m0 ? mean0 : T(0))) * r * eqw[u];
    if (f < m0) val += eqb[f];
    xhat_bt[bt.at(n, u, m, 0, 0)] = val;
  }
  if (lane == 0) {
    stats[4 * n] = mean;
    stats[4 * n + 1] = rstd;
    stats[4 * n + 2] = mean0;
    stats[4 * n + 3] = r;
  }
}

// g_s = res_s + LN^T(g_shat), g_x = res_x + EqLN^T(g_xhat_bt)   (res_* may be NULL)
template <typename T>
__global__ void k_norm_bwd(const T* __restrict__ s, const T* __restrict__ x, const T* __restrict__ lnw,
                           const T* __restrict__ eqw, const T* __restrict__ stats, int64_t N, int F, Irreps ir,
                           int do_norm, const T* __restrict__ g_shat, int64_t ld_gs, const T* __restrict__ g_xhat_bt,
                           const T* __restrict__ res_s, const T* __restrict__ res_x, T* __restrict__ g_s,
                           T* __restrict__ g_x) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (n >= N) return;
  BT bt{ir, N};
  if (!do_norm) {
    for (int f = lane; f < F; f += 64) g_s[n * F + f] = g_shat[n * ld_gs + f] + (res_s ? res_s[n * F + f] : T(0));
    for (int f = lane; f < D; f += 64) {
      int u, m;
      chan_of_flat(ir, f, u, m);
      g_x[n * D + f] = g_xhat_bt[bt.at(n, u, m, 0, 0)] + (res_x ? res_x[n * D + f] : T(0));
    }
    return;
  }
  const T mean = stats[4 * n], rstd = stats[4 * n + 1], mean0 = stats[4 * n + 2], r = stats[4 * n + 3];
  // LayerNorm backward: yhat = (s-mean) rstd, dy = g w: g_s = rstd (dy - mean(dy) - yhat mean(dy yhat))
  T a1 = T(0), a2 = T(0);
  for (int f = lane; f < F; f += 64) {
    T yh = (s[n * F + f] - mean) * rstd, dy = g_shat[n * ld_gs + f] * lnw[f];
    a1 += dy;
    a2 += dy * yh;
  }
  a1 = wave_sum<T>(a1) / T(F);
  a2 = wave_sum<T>(a2) / T(F);
  for (int f = lane; f < F; f += 64) {
    T yh = (s[n * F + f] - mean) * rstd, dy = g_shat[n * ld_gs + f] * lnw[f];
    g_s[n * F + f] = rstd * (dy - a1 - yh * a2) + (res_s ? res_s[n * F + f] : T(0));
  }
  // EquivariantLayerNorm backward
  const T* xr = x + n * D;
  T dotp = T(0);
  for (int f = lane; f < D; f += 64) {
    int u, m;
    chan_of_flat(ir, f, u, m);
    T xc = xr[f] - (f < m0 ? mean0 : T(0));
    dotp += g_xhat_bt[bt.at(n, u, m, 0, 0)] * eqw[u] * xc;
  }
  const T coef = wave_sum<T>(dotp) * r * r * r / T(C);
  T gs = T(0);
  for (int f = lane; f < m0; f += 64) gs += r * g_xhat_bt[bt.at(n, f, 0, 0, 0)] * eqw[f] - coef * (xr[f] - mean0);
  const T gmean = m0 > 0 ? wave_sum<T>(gs) / T(m0) : T(0);
  for (int f = lane; f < D; f += 64) {
    int u, m;
    chan_of_flat(ir, f, u, m);
    T xc = xr[f] - (f < m0 ? mean0 : T(0));
    T val = r * g_xhat_bt[bt.at(n, u, m, 0, 0)] * eqw[u] - coef * xc;
    if (f < m0) val -= gmean;
    g_x[n * D + f] = val + (res_x ? res_x[n * D + f] : T(0));
  }
}

// ---- v = |V|, p = <U,V> per channel (Invariant nn/o3layer.py:39-44, EquivariantDot :104-109) ----
// one thread per (node, channel); writes v into cat[n, F + u] (row stride ld_cat) and p[n, u]
template <typename T>
__global__ void k_uv_reduce_fwd(const T* __restrict__ uv_bt, int64_t N, Irreps ir, T eps, T* __restrict__ cat,
                                int64_t ld_cat, int F, T* __restrict__ p) {
  const int C = ir.C();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * C) return;
  const int64_t n = t / C;
  const int u = (int)(t - n * C);
  BT bt{ir, N};
  int l, off;
  ir.locate(u, l, off);
  T vv = T(0), uv = T(0);
  for (int m = 0; m < 2 * l + 1; ++m) {
    T U = uv_bt[bt.at(n, u, m, 1, 0)], V = uv_bt[bt.at(n, u, m, 1, 1)];
    vv += V * V;
    uv += U * V;
  }
  cat[n * ld_cat + F + u] = sqrt_<T>(vv + eps * eps) - eps;
  p[t] = uv;
}

// g_U = dL/dU(output stage) + g_p V ;  g_V = g_p U + g_v V / (v + eps).  The output stage's part is either already in
// the U columns of g_uv_bt (g_x_out == NULL: read-modify-write) or formed here as g_x_out[n, u, m] a_vv[n, u].
template <typename T>
__global__ void k_uv_reduce_bwd(const T* __restrict__ uv_bt, const T* __restrict__ g_p, const T* __restrict__ g_cat,
                                int64_t ld_cat, int F, int64_t N, Irreps ir, T eps, const T* __restrict__ g_x_out,
                                const T* __restrict__ a, T* __restrict__ g_uv_bt) {
  const int C = ir.C(), D = ir.D(), A = C + 2 * F;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * C) return;
  const int64_t n = t / C;
  const int u = (int)(t - n * C);
  BT bt{ir, N};
  int l, off;
  ir.locate(u, l, off);
  T vv = T(0);
  for (int m = 0; m < 2 * l + 1; ++m) { T V = uv_bt[bt.at(n, u, m, 1, 1)]; vv += V * V; }
  const T gp = g_p[t];
  const T gv = g_cat[n * ld_cat + F + u] / sqrt_<T>(vv + eps * eps);
  for (int m = 0; m < 2 * l + 1; ++m) {
    const int64_t iu = bt.at(n, u, m, 1, 0), iv = bt.at(n, u, m, 1, 1);
    T U = uv_bt[iu], V = uv_bt[iv];
    const T gu = g_x_out ? g_x_out[n * D + off + m] * a[n * A + u] : g_uv_bt[iu];
    g_uv_bt[iu] = gu + gp * V;
    g_uv_bt[iv] = gp * U + gv * V;
  }
}

// ---- update output stage (nn/xpainn.py:218-229): a = [a_vv C | a_sv F | a_ss F] --------------
//   s_out = s + a_sv * ip + a_ss ;  x_out[n,u,m] = x + U[n,u,m] * a_vv[u]      (x in e3nn layout)
template <typename T>
__global__ void k_update_out_fwd(const T* __restrict__ s, const T* __restrict__ x, const T* __restrict__ uv_bt,
                                 const T* __restrict__ a, const T* __restrict__ ip, int64_t N, int F, Irreps ir,
                                 T* __restrict__ s_out, T* __restrict__ x_out) {
  const int D = ir.D(), C = ir.C(), W = F + D, A = C + 2 * F;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * W) return;
  const int64_t n = t / W;
  const int f = (int)(t - n * W);
  if (f < F) {
    s_out[n * F + f] = s[n * F + f] + a[n * A + C + f] * ip[n * F + f] + a[n * A + C + F + f];
  } else {
    const int fx = f - F;
    int u, m;
    chan_of_flat(ir, fx, u, m);
    BT bt{ir, N};
    x_out[n * D + fx] = x[n * D + fx] + uv_bt[bt.at(n, u, m, 1, 0)] * a[n * A + u];
  }
}

// g_a = [sum_m U g_x | g_s ip | g_s], g_ip = g_s a_sv, g_U = g_x a_vv (written into the U columns of g_uv_bt)
template <typename T>
__global__ void k_update_out_bwd(const T* __restrict__ g_s_out, const T* __restrict__ g_x_out,
                                 const T* __restrict__ uv_bt, const T* __restrict__ a, const T* __restrict__ ip,
                                 int64_t N, int F, Irreps ir, T* __restrict__ g_a, T* __restrict__ g_ip,
                                 T* __restrict__ g_uv_bt) {
  const int D = ir.D(), C = ir.C(), A = C + 2 * F, W = C + F;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * W) return;
  const int64_t n = t / W;
  const int f = (int)(t - n * W);
  if (f < C) {  // gate channel u = f
    BT bt{ir, N};
    int l, off;
    ir.locate(f, l, off);
    const T avv = a[n * A + f];
    T acc = T(0);
    for (int m = 0; m < 2 * l + 1; ++m) {
      const T gx = g_x_out[n * D + off + m];
      const int64_t iu = bt.at(n, f, m, 1, 0);
      acc += uv_bt[iu] * gx;
      if (g_uv_bt) g_uv_bt[iu] = gx * avv;
    }
    g_a[n * A + f] = acc;
  } else {  // scalar channel
    const int c = f - C;
    const T gs = g_s_out[n * F + c];
    g_a[n * A + C + c] = gs * ip[n * F + c];
    g_a[n * A + C + F + c] = gs;
    g_ip[n * F + c] = gs * a[n * A + C + c];
  }
}

// =================================================================================================
// Fast paths.  The kernels above index one element per thread with 64-bit divisions by run-time row
// widths; the ones below give a thread a fixed COLUMN (its channel/component decoding, BT offsets
// and weights are computed once) and walk NPB consecutive nodes with all loads of a node issued
// together, and the norms keep a node's row in registers (one wave per node, DPP reductions).
// =================================================================================================
constexpr int NODE_NPB = 8;  // nodes per workgroup of the column kernels ...
// ... and 1 when that would leave most of the chip idle (MD-sized systems: a 192-atom box is 24 workgroups at 8)
static inline int node_npb(int64_t n) { return n >= 8 * 1024 ? NODE_NPB : 1; }
// workgroup width (a multiple of 64 in [256, 512]) that covers `cols` columns with the fewest idle lanes: the default
// model's output stage has 608 / 352 columns, i.e. 2 x 320 and 1 x 384 instead of 3 x 256 and 2 x 256
static inline int node_block(int cols) {
  int best = 256, waste = (cols + 255) / 256 * 256 - cols;
  for (int bs = 320; bs <= 512; bs += 64) {
    const int w = (cols + bs - 1) / bs * bs - cols;
    if (w < waste) {
      waste = w;
      best = bs;
    }
  }
  return best;
}

__device__ __forceinline__ float wave_total_n(float v) {
#define XEQ_N_DPP(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xF, true)))
  v = XEQ_N_DPP(v, 0xB1, 0xF);
  v = XEQ_N_DPP(v, 0x4E, 0xF);
  v = XEQ_N_DPP(v, 0x141, 0xF);
  v = XEQ_N_DPP(v, 0x140, 0xF);
  v = XEQ_N_DPP(v, 0x142, 0xA);
  v = XEQ_N_DPP(v, 0x143, 0xC);
#undef XEQ_N_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_total_n(double v) { return wave_sum<double>(v); }

// per-thread description of gate channel u in the BT buffers
struct ChanBT {
  int l, d, up;
  int64_t base_x, base_uv;  // element offset of (n = 0, m = 0): xhat-like (width mul) / U|V pair (width 2 mul)
  int w;                    // mul_l
  int off;                  // flat e3nn offset of component 0
};
__device__ __forceinline__ ChanBT chan_bt(const Irreps& ir, int64_t N, int u) {
  ChanBT c;
  ir.locate(u, c.l, c.off);
  c.d = 2 * c.l + 1;
  c.w = ir.mul[c.l];
  c.up = u - (c.l == 0 ? 0 : (c.l == 1 ? ir.mul[0] : ir.mul[0] + ir.mul[1]));
  const int64_t b = c.l == 0 ? 0 : (c.l == 1 ? (int64_t)ir.mul[0] : (int64_t)ir.mul[0] + 3 * ir.mul[1]);
  c.base_x = N * b + c.up;
  c.base_uv = 2 * N * b + c.up;
  return c;
}

// v = |V|, p = <U,V>: thread = channel, NPB nodes per workgroup
template <typename T, int NPB>
__global__ void __launch_bounds__(256) k_uv_reduce_fwd_c(const T* __restrict__ uv_bt, int64_t N, Irreps ir, T eps,
                                                          T* __restrict__ cat, int64_t ld_cat, int F, T* __restrict__ p) {
  const int C = ir.C(), u = threadIdx.x;
  if (u >= C) return;
  const ChanBT c = chan_bt(ir, N, u);
  const int64_t n0 = (int64_t)blockIdx.x * NPB;
  const int w2 = 2 * c.w;
#ifndef XEQ_UVF_UNROLL
#define XEQ_UVF_UNROLL 4
#endif
  constexpr int UNROLL = XEQ_UVF_UNROLL;
#pragma unroll UNROLL
  for (int j = 0; j < NPB; ++j) {
    const int64_t n = n0 + j;
    if (n >= N) break;
    const T* row = uv_bt + c.base_uv + n * c.d * w2;
    T vv = T(0), uv = T(0);
    for (int m = 0; m < c.d; ++m) {
      const T U = row[m * w2], V = row[m * w2 + c.w];
      vv += V * V;
      uv += U * V;
    }
    cat[n * ld_cat + F + u] = sqrt_<T>(vv + eps * eps) - eps;
    p[n * C + u] = uv;
  }
}

template <typename T, int NPB>
__global__ void __launch_bounds__(256) k_uv_reduce_bwd_c(const T* __restrict__ uv_bt, const T* __restrict__ g_p,
                                                          const T* __restrict__ g_cat, int64_t ld_cat, int F, int64_t N,
                                                          Irreps ir, T eps, const T* __restrict__ g_x_out,
                                                          const T* __restrict__ a, T* __restrict__ g_uv_bt) {
  const int C = ir.C(), D = ir.D(), A = C + 2 * F, u = threadIdx.x;
  if (u >= C) return;
  const ChanBT c = chan_bt(ir, N, u);
  const int64_t n0 = (int64_t)blockIdx.x * NPB;
  const int w2 = 2 * c.w;
  // measured (scratch/run_node_variants.sh, 18.6 k nodes): unroll 1 / 2 / 4 / 8 = 59 / 84 / 117 / 50 us at 90 / 133 / 230 / 52
  // VGPRs: what the scheduler does with the unrolled loads decides the occupancy, and with it the time
#ifndef XEQ_UVB_UNROLL
#define XEQ_UVB_UNROLL 8
#endif
  constexpr int UNROLL = XEQ_UVB_UNROLL;
#pragma unroll UNROLL
  for (int j = 0; j < NPB; ++j) {
    const int64_t n = n0 + j;
    if (n >= N) break;
    const int64_t r0 = c.base_uv + n * c.d * w2;
    const T gp = g_p[n * C + u], gc = g_cat[n * ld_cat + F + u];
    const T avv = g_x_out ? a[n * A + u] : T(0);
    T U[5], V[5], GU[5];
    T vv = T(0);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      if (m < c.d) {
        U[m] = uv_bt[r0 + m * w2];
        V[m] = uv_bt[r0 + m * w2 + c.w];
        GU[m] = g_x_out ? g_x_out[n * D + c.off + m] * avv : g_uv_bt[r0 + m * w2];
        vv += V[m] * V[m];
      }
    }
    const T gv = gc / sqrt_<T>(vv + eps * eps);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      if (m < c.d) {
        g_uv_bt[r0 + m * w2] = GU[m] + gp * V[m];
        g_uv_bt[r0 + m * w2 + c.w] = gp * U[m] + gv * V[m];
      }
    }
  }
}

// update output stage: thread = flat column of [s (F) | x (D)], column blocks of 256
template <typename T, int NPB>
__global__ void __launch_bounds__(512) k_update_out_fwd_c(const T* __restrict__ s, const T* __restrict__ x,
                                                           const T* __restrict__ uv_bt, const T* __restrict__ a,
                                                           const T* __restrict__ ip, int64_t N, int F, Irreps ir,
                                                           int ncb, T* __restrict__ s_out, T* __restrict__ x_out) {
  const int D = ir.D(), C = ir.C(), A = C + 2 * F;
  const int cb = blockIdx.x % ncb;
  const int64_t n0 = (int64_t)(blockIdx.x / ncb) * NPB;
  const int f = cb * (int)blockDim.x + threadIdx.x;
  if (f >= F + D) return;
  if (f < F) {
#pragma unroll 4
    for (int j = 0; j < NPB; ++j) {
      const int64_t n = n0 + j;
      if (n >= N) break;
      s_out[n * F + f] = s[n * F + f] + a[n * A + C + f] * ip[n * F + f] + a[n * A + C + F + f];
    }
  } else if (x_out != nullptr) {   // (NULL: nobody reads the block's equivariant output -- the last block in front of a scalar head)
    const int fx = f - F;
    int u, m;
    chan_of_flat(ir, fx, u, m);
    const ChanBT c = chan_bt(ir, N, u);
    const int w2 = 2 * c.w;
    const int64_t ub = c.base_uv + (int64_t)m * w2;
#ifndef XEQ_UOF_UNROLL
#define XEQ_UOF_UNROLL 4
#endif
    constexpr int UNROLL = XEQ_UOF_UNROLL;
#pragma unroll UNROLL
    for (int j = 0; j < NPB; ++j) {
      const int64_t n = n0 + j;
      if (n >= N) break;
      x_out[n * D + fx] = x[n * D + fx] + uv_bt[ub + n * c.d * w2] * a[n * A + u];
    }
  }
}

// thread = column of [gate channels (C) | scalar channels (F)]
template <typename T, int NPB>
__global__ void __launch_bounds__(512) k_update_out_bwd_c(const T* __restrict__ g_s_out, const T* __restrict__ g_x_out,
                                                           const T* __restrict__ uv_bt, const T* __restrict__ a,
                                                           const T* __restrict__ ip, int64_t N, int F, Irreps ir, int ncb,
                                                           T* __restrict__ g_a, T* __restrict__ g_ip,
                                                           T* __restrict__ g_uv_bt) {
  const int D = ir.D(), C = ir.C(), A = C + 2 * F;
  const int cb = blockIdx.x % ncb;
  const int64_t n0 = (int64_t)(blockIdx.x / ncb) * NPB;
  const int f = cb * (int)blockDim.x + threadIdx.x;
  if (f >= C + F) return;
  if (f < C && g_x_out == nullptr) {   // the block's equivariant output has no consumer (last block of a force evaluation): zeros
    for (int j = 0; j < NPB; ++j) {
      const int64_t n = n0 + j;
      if (n >= N) break;
      g_a[n * A + f] = T(0);
    }
  } else if (f < C) {
    const ChanBT c = chan_bt(ir, N, f);
    const int w2 = 2 * c.w;
#ifndef XEQ_UOB_UNROLL
#define XEQ_UOB_UNROLL 8   // 41 us against 45 (unroll 2) and 62 (unroll 4: 104 VGPRs)
#endif
    constexpr int UNROLL = XEQ_UOB_UNROLL;
#pragma unroll UNROLL
    for (int j = 0; j < NPB; ++j) {
      const int64_t n = n0 + j;
      if (n >= N) break;
      const int64_t r0 = c.base_uv + n * c.d * w2;
      const T avv = a[n * A + f];
      T gx[5], U[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        if (m < c.d) {
          gx[m] = g_x_out[n * D + c.off + m];
          U[m] = uv_bt[r0 + m * w2];
        }
      }
      T acc = T(0);
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        if (m < c.d) {
          acc += U[m] * gx[m];
          if (g_uv_bt) g_uv_bt[r0 + m * w2] = gx[m] * avv;
        }
      }
      g_a[n * A + f] = acc;
    }
  } else {
    const int cc = f - C;
#pragma unroll 4
    for (int j = 0; j < NPB; ++j) {
      const int64_t n = n0 + j;
      if (n >= N) break;
      const T gs = g_s_out[n * F + cc];
      g_a[n * A + C + cc] = gs * ip[n * F + cc];
      g_a[n * A + C + F + cc] = gs;
      g_ip[n * F + cc] = gs * a[n * A + C + cc];
    }
  }
}

// ---- norms with the node's row in registers: one wave per node, SS scalar slots and XS equivariant slots per lane
template <typename T, int SS, int XS>
__global__ void __launch_bounds__(256) k_norm_fwd_r(const T* __restrict__ s, const T* __restrict__ x,
                                                     const T* __restrict__ lnw, const T* __restrict__ lnb,
                                                     const T* __restrict__ eqw, const T* __restrict__ eqb, int64_t N, int F,
                                                     Irreps ir, T* __restrict__ shat, int64_t ld_s,
                                                     T* __restrict__ xhat_bt, T* __restrict__ stats) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // slot decoding, once per wave
  T w_s[SS], b_s[SS], w_x[XS], b_x[XS];
  uint32_t o_x[XS], st_x[XS];  // BT element offset of (n = 0) and node stride: the launcher checks N * D < 2^31
#pragma unroll
  for (int k = 0; k < SS; ++k) {
    const int f = lane + 64 * k;
    w_s[k] = f < F ? lnw[f] : T(0);
    b_s[k] = f < F ? lnb[f] : T(0);
  }
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    const int f = lane + 64 * k;
    int u = 0, m = 0;
    if (f < D) chan_of_flat(ir, f, u, m);
    const ChanBT c = chan_bt(ir, N, u);
    w_x[k] = f < D ? eqw[u] : T(0);
    b_x[k] = f < m0 ? eqb[f] : T(0);
    o_x[k] = (uint32_t)(c.base_x + (int64_t)m * c.w);
    st_x[k] = (uint32_t)(c.d * c.w);
  }
  for (int64_t n = wid; n < N; n += nw) {
    T sv[SS], xv[XS];
#pragma unroll
    for (int k = 0; k < SS; ++k) sv[k] = (lane + 64 * k < F) ? s[n * F + lane + 64 * k] : T(0);
#pragma unroll
    for (int k = 0; k < XS; ++k) xv[k] = (lane + 64 * k < D) ? x[n * D + lane + 64 * k] : T(0);
    T a = T(0), q = T(0);
#pragma unroll
    for (int k = 0; k < SS; ++k) a += sv[k];
#pragma unroll
    for (int k = 0; k < XS; ++k) q += (lane + 64 * k < m0) ? xv[k] : T(0);
    const T mean = wave_total_n(a) / T(F);
    const T mean0 = m0 > 0 ? wave_total_n(q) / T(m0) : T(0);
    T v = T(0), sq = T(0);
#pragma unroll
    for (int k = 0; k < SS; ++k) {
      const T d = (lane + 64 * k < F) ? sv[k] - mean : T(0);
      sv[k] = d;
      v += d * d;
    }
#pragma unroll
    for (int k = 0; k < XS; ++k) {
      const int f = lane + 64 * k;
      const T d = f < D ? xv[k] - (f < m0 ? mean0 : T(0)) : T(0);
      xv[k] = d;
      sq += d * d;
    }
    const T rstd = T(1) / sqrt_<T>(wave_total_n(v) / T(F) + T(1e-5));
    const T r = T(1) / sqrt_<T>(wave_total_n(sq) / T(C) + T(1e-5));
#pragma unroll
    for (int k = 0; k < SS; ++k)
      if (lane + 64 * k < F) shat[n * ld_s + lane + 64 * k] = sv[k] * rstd * w_s[k] + b_s[k];
#pragma unroll
    for (int k = 0; k < XS; ++k)
      if (lane + 64 * k < D) xhat_bt[o_x[k] + (uint32_t)n * st_x[k]] = xv[k] * r * w_x[k] + b_x[k];
    if (lane == 0) {
      stats[4 * n] = mean;
      stats[4 * n + 1] = rstd;
      stats[4 * n + 2] = mean0;
      stats[4 * n + 3] = r;
    }
  }
}

template <typename T, int SS, int XS>
__global__ void __launch_bounds__(256) k_norm_bwd_r(const T* __restrict__ s, const T* __restrict__ x,
                                                     const T* __restrict__ lnw, const T* __restrict__ eqw,
                                                     const T* __restrict__ stats, int64_t N, int F, Irreps ir,
                                                     const T* __restrict__ g_shat, int64_t ld_gs,
                                                     const T* __restrict__ g_xhat_bt, const T* __restrict__ res_s,
                                                     const T* __restrict__ res_x, T* __restrict__ g_s, T* __restrict__ g_x) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  T w_s[SS], w_x[XS];
  int64_t o_x[XS];
  int st_x[XS];
#pragma unroll
  for (int k = 0; k < SS; ++k) w_s[k] = (lane + 64 * k < F) ? lnw[lane + 64 * k] : T(0);
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    const int f = lane + 64 * k;
    int u = 0, m = 0;
    if (f < D) chan_of_flat(ir, f, u, m);
    const ChanBT c = chan_bt(ir, N, u);
    w_x[k] = f < D ? eqw[u] : T(0);
    o_x[k] = c.base_x + (int64_t)m * c.w;
    st_x[k] = c.d * c.w;
  }
  for (int64_t n = wid; n < N; n += nw) {
    const T mean = stats[4 * n], rstd = stats[4 * n + 1], mean0 = stats[4 * n + 2], r = stats[4 * n + 3];
    T yh[SS], dy[SS], rs[SS], xc[XS], gw[XS], rx[XS];
#pragma unroll
    for (int k = 0; k < SS; ++k) {
      const int f = lane + 64 * k;
      const bool ok = f < F;
      yh[k] = ok ? (s[n * F + f] - mean) * rstd : T(0);
      dy[k] = ok ? g_shat[n * ld_gs + f] * w_s[k] : T(0);
      rs[k] = (ok && res_s) ? res_s[n * F + f] : T(0);
    }
#pragma unroll
    for (int k = 0; k < XS; ++k) {
      const int f = lane + 64 * k;
      const bool ok = f < D;
      xc[k] = ok ? x[n * D + f] - (f < m0 ? mean0 : T(0)) : T(0);
      gw[k] = ok ? g_xhat_bt[o_x[k] + n * st_x[k]] * w_x[k] : T(0);
      rx[k] = (ok && res_x) ? res_x[n * D + f] : T(0);
    }
    T a1 = T(0), a2 = T(0), dotp = T(0);
#pragma unroll
    for (int k = 0; k < SS; ++k) {
      a1 += dy[k];
      a2 += dy[k] * yh[k];
    }
#pragma unroll
    for (int k = 0; k < XS; ++k) dotp += gw[k] * xc[k];
    a1 = wave_total_n(a1) / T(F);
    a2 = wave_total_n(a2) / T(F);
    const T coef = wave_total_n(dotp) * r * r * r / T(C);
    T gs = T(0);
#pragma unroll
    for (int k = 0; k < XS; ++k) gs += (lane + 64 * k < m0) ? r * gw[k] - coef * xc[k] : T(0);
    const T gmean = m0 > 0 ? wave_total_n(gs) / T(m0) : T(0);
#pragma unroll
    for (int k = 0; k < SS; ++k)
      if (lane + 64 * k < F) g_s[n * F + lane + 64 * k] = rstd * (dy[k] - a1 - yh[k] * a2) + rs[k];
#pragma unroll
    for (int k = 0; k < XS; ++k) {
      const int f = lane + 64 * k;
      if (f < D) g_x[n * D + f] = r * gw[k] - coef * xc[k] - (f < m0 ? gmean : T(0)) + rx[k];
    }
  }
}

// Reverse of the two norms with ONE reduction stage: the five wave sums a node needs (sum dy, sum dy yhat, sum gw xc and,
// for the 0e mean, sum gw and sum xc over the scalar channels) do not depend on each other once the last one is
// split as  sum_{0e} (r gw - coef xc) = r sum gw - coef sum xc,  so every load of the node is issued up front, the
// sums run interleaved, and the node's row stays in registers for the output (one memory round trip per node
// instead of four dependent ones).  NPW nodes per wave amortise the slot decoding.
template <typename T, int SS, int XS>
__global__ void __launch_bounds__(256) k_norm_bwd_f(const T* __restrict__ s, const T* __restrict__ x,
                                                     const T* __restrict__ lnw, const T* __restrict__ eqw,
                                                     const T* __restrict__ stats, int64_t N, int F, Irreps ir,
                                                     const T* __restrict__ g_shat, int64_t ld_gs,
                                                     const T* __restrict__ g_xhat_bt, const T* __restrict__ res_s,
                                                     const T* __restrict__ res_x, T* __restrict__ g_s, T* __restrict__ g_x) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  T w_s[SS], w_x[XS];
  uint32_t o_x[XS], st_x[XS];   // BT element offsets: the launcher checks N * D < 2^31
#pragma unroll
  for (int k = 0; k < SS; ++k) w_s[k] = (lane + 64 * k < F) ? lnw[lane + 64 * k] : T(0);
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    const int f = lane + 64 * k;
    int u = 0, m = 0;
    if (f < D) chan_of_flat(ir, f, u, m);
    const ChanBT c = chan_bt(ir, N, u);
    w_x[k] = f < D ? eqw[u] : T(0);
    o_x[k] = (uint32_t)(c.base_x + (int64_t)m * c.w);
    st_x[k] = (uint32_t)(c.d * c.w);
  }
  for (int64_t n = wid; n < N; n += nw) {
    T yh[SS], dy[SS], xc[XS], gw[XS];
    T rs[SS], rx[XS];   // residual rows ride along with the first (only) round of loads: 58 -> 41 -> 33 us with them
#pragma unroll
    for (int k = 0; k < SS; ++k) rs[k] = (res_s && lane + 64 * k < F) ? res_s[n * F + lane + 64 * k] : T(0);
#pragma unroll
    for (int k = 0; k < XS; ++k) rx[k] = (res_x && lane + 64 * k < D) ? res_x[n * D + lane + 64 * k] : T(0);
    const T mean = stats[4 * n], rstd = stats[4 * n + 1], mean0 = stats[4 * n + 2], r = stats[4 * n + 3];
#pragma unroll
    for (int k = 0; k < SS; ++k) {
      const int f = lane + 64 * k;
      const bool ok = f < F;
      yh[k] = ok ? s[n * F + f] : T(0);
      dy[k] = ok ? g_shat[n * ld_gs + f] : T(0);
    }
#pragma unroll
    for (int k = 0; k < XS; ++k) {
      const int f = lane + 64 * k;
      const bool ok = f < D;
      xc[k] = ok ? x[n * D + f] : T(0);
      gw[k] = ok ? g_xhat_bt[o_x[k] + (uint32_t)n * st_x[k]] : T(0);
    }
    T a1 = T(0), a2 = T(0), dotp = T(0), sg0 = T(0), sx0 = T(0);
#pragma unroll
    for (int k = 0; k < SS; ++k) {
      const bool ok = lane + 64 * k < F;
      yh[k] = ok ? (yh[k] - mean) * rstd : T(0);
      dy[k] *= w_s[k];
      a1 += dy[k];
      a2 += dy[k] * yh[k];
    }
#pragma unroll
    for (int k = 0; k < XS; ++k) {
      const int f = lane + 64 * k;
      const bool z = f < m0;
      xc[k] = f < D ? xc[k] - (z ? mean0 : T(0)) : T(0);
      gw[k] *= w_x[k];
      dotp += gw[k] * xc[k];
      sg0 += z ? gw[k] : T(0);
      sx0 += z ? xc[k] : T(0);
    }
    a1 = wave_total_n(a1);
    a2 = wave_total_n(a2);
    dotp = wave_total_n(dotp);
    sg0 = wave_total_n(sg0);
    sx0 = wave_total_n(sx0);
    a1 /= T(F);
    a2 /= T(F);
    const T coef = dotp * r * r * r / T(C);
    const T gmean = m0 > 0 ? (r * sg0 - coef * sx0) / T(m0) : T(0);
#pragma unroll
    for (int k = 0; k < SS; ++k) {
      const int f = lane + 64 * k;
      if (f < F) g_s[n * F + f] = rstd * (dy[k] - a1 - yh[k] * a2) + rs[k];
    }
#pragma unroll
    for (int k = 0; k < XS; ++k) {
      const int f = lane + 64 * k;
      if (f < D) g_x[n * D + f] = r * gw[k] - coef * xc[k] - (f < m0 ? gmean : T(0)) + rx[k];
    }
  }
}

static inline int irreps_from(const int32_t mul[3], Irreps& ir, const char* who) {
  for (int l = 0; l < 3; ++l) {
    if (mul[l] < 0) {
      set_error("%s: negative multiplicity", who);
      return XEQ_ERR_INVALID_ARGUMENT;
    }
    ir.mul[l] = mul[l];
  }
  if (ir.C() == 0) {
    set_error("%s: empty irreps", who);
    return XEQ_ERR_INVALID_ARGUMENT;
  }
  return XEQ_OK;
}

}  // namespace xeq

using namespace xeq;

#define XEQ_IR(who)                        \
  Irreps ir;                               \
  {                                        \
    int rc_ = irreps_from(mul, ir, who);   \
    if (rc_ != XEQ_OK) return rc_;         \
  }

extern "C" {

int xeq_norm_fwd(int dtype, const void* s, const void* x, const void* ln_w, const void* ln_b, const void* eq_w,
                 const void* eq_b, int64_t n, int node_dim, const int32_t mul[3], int do_norm, void* shat, int64_t ld_s,
                 void* xhat_bt, void* stats, void* stream) {
  XEQ_IR("xeq_norm_fwd");
  XEQ_CHECK_ARG(node_dim > 0 && ld_s >= node_dim, "xeq_norm_fwd: bad node_dim / row stride");
  if (n <= 0) return XEQ_OK;
  const bool fast = do_norm && node_dim <= 128 && ir.D() <= 512 && n * (int64_t)ir.D() < (1ll << 31);  // row in registers: 2 + 8 slots per lane
  // ~4 nodes per wave: the slot decoding of a wave is amortised, every CU still gets >= 16 waves at 18k nodes
  // (small systems: one node per wave, so that a 192-atom box still spreads over 48 workgroups)
#ifndef XEQ_NORM_FWD_NPW
#define XEQ_NORM_FWD_NPW 4
#endif
  const int64_t per_wg = n >= 8 * 1024 ? 4 * XEQ_NORM_FWD_NPW : 4;
  const unsigned wgrid = (unsigned)((n + per_wg - 1) / per_wg < 65536 ? (n + per_wg - 1) / per_wg : 65536);
  XEQ_DISPATCH_FLOAT(dtype, {
    if (fast)
      hipLaunchKernelGGL((k_norm_fwd_r<T, 2, 8>), dim3(wgrid), dim3(256), 0, (hipStream_t)stream, (const T*)s, (const T*)x,
                         (const T*)ln_w, (const T*)ln_b, (const T*)eq_w, (const T*)eq_b, n, node_dim, ir, (T*)shat, ld_s,
                         (T*)xhat_bt, (T*)stats);
    else
      hipLaunchKernelGGL((k_norm_fwd<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)s,
                         (const T*)x, (const T*)ln_w, (const T*)ln_b, (const T*)eq_w, (const T*)eq_b, n, node_dim, ir,
                         do_norm, (T*)shat, ld_s, (T*)xhat_bt, (T*)stats);
  });
  XEQ_CHECK_LAUNCH("xeq_norm_fwd");
  return XEQ_OK;
}

int xeq_norm_bwd(int dtype, const void* s, const void* x, const void* ln_w, const void* eq_w, const void* stats,
                 int64_t n, int node_dim, const int32_t mul[3], int do_norm, const void* g_shat, int64_t ld_gs,
                 const void* g_xhat_bt, const void* res_s, const void* res_x, void* g_s, void* g_x, void* stream) {
  XEQ_IR("xeq_norm_bwd");
  XEQ_CHECK_ARG(node_dim > 0 && ld_gs >= node_dim, "xeq_norm_bwd: bad node_dim / row stride");
  if (n <= 0) return XEQ_OK;
  // measured (QM9-1024, scratch/run_node_variants.sh): element-per-lane kernel with four dependent passes 58 us; row in
  // registers with four dependent reductions 76-93 us; row in registers with one fused reduction stage: see k_norm_bwd_f
#ifndef XEQ_NORM_BWD_NPW
#define XEQ_NORM_BWD_NPW 4
#endif
  const bool fast = do_norm && node_dim <= 128 && ir.D() <= 512 && n * (int64_t)ir.D() < (1ll << 31) &&
                    getenv("XEQ_NORM_BWD_OLD") == nullptr;
  const int64_t per_wg = n >= 8 * 1024 ? 4 * XEQ_NORM_BWD_NPW : 4;
  const unsigned wgrid = (unsigned)((n + per_wg - 1) / per_wg);
  XEQ_DISPATCH_FLOAT(dtype, {
    if (fast)
      hipLaunchKernelGGL((k_norm_bwd_f<T, 2, 8>), dim3(wgrid), dim3(256), 0, (hipStream_t)stream, (const T*)s, (const T*)x,
                         (const T*)ln_w, (const T*)eq_w, (const T*)stats, n, node_dim, ir, (const T*)g_shat, ld_gs,
                         (const T*)g_xhat_bt, (const T*)res_s, (const T*)res_x, (T*)g_s, (T*)g_x);
    else
      hipLaunchKernelGGL((k_norm_bwd<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)s,
                         (const T*)x, (const T*)ln_w, (const T*)eq_w, (const T*)stats, n, node_dim, ir, do_norm,
                         (const T*)g_shat, ld_gs, (const T*)g_xhat_bt, (const T*)res_s, (const T*)res_x, (T*)g_s,
                         (T*)g_x);
  });
  XEQ_CHECK_LAUNCH("xeq_norm_bwd");
  return XEQ_OK;
}

int xeq_uv_reduce_fwd(int dtype, const void* uv_bt, int64_t n, const int32_t mul[3], double eps, void* cat,
                      int64_t ld_cat, int node_dim, void* p, void* stream) {
  XEQ_IR("xeq_uv_reduce_fwd");
  if (n <= 0) return XEQ_OK;
  const int64_t total = n * ir.C();
  XEQ_DISPATCH_FLOAT(dtype, {
    if (ir.C() <= 256)
    {
      if (node_npb(n) == 1)
        hipLaunchKernelGGL((k_uv_reduce_fwd_c<T, 1>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const T*)uv_bt, n,
                           ir, (T)eps, (T*)cat, ld_cat, node_dim, (T*)p);
      else
        hipLaunchKernelGGL((k_uv_reduce_fwd_c<T, NODE_NPB>), dim3((unsigned)((n + NODE_NPB - 1) / NODE_NPB)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)uv_bt, n, ir, (T)eps, (T*)cat, ld_cat, node_dim, (T*)p);
    }
    else
      hipLaunchKernelGGL((k_uv_reduce_fwd<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)uv_bt, n, ir, (T)eps, (T*)cat, ld_cat, node_dim, (T*)p);
  });
  XEQ_CHECK_LAUNCH("xeq_uv_reduce_fwd");
  return XEQ_OK;
}

int xeq_uv_reduce_bwd(int dtype, const void* uv_bt, const void* g_p, const void* g_cat, int64_t ld_cat, int node_dim,
                      int64_t n, const int32_t mul[3], double eps, const void* g_x_out, const void* a, void* g_uv_bt,
                      void* stream) {
  XEQ_IR("xeq_uv_reduce_bwd");
  XEQ_CHECK_ARG((g_x_out == nullptr) == (a == nullptr), "xeq_uv_reduce_bwd: g_x_out and a go together");
  if (n <= 0) return XEQ_OK;
  const int64_t total = n * ir.C();
  XEQ_DISPATCH_FLOAT(dtype, {
    if (ir.C() <= 256) {
      if (node_npb(n) == 1)
        hipLaunchKernelGGL((k_uv_reduce_bwd_c<T, 1>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const T*)uv_bt,
                           (const T*)g_p, (const T*)g_cat, ld_cat, node_dim, n, ir, (T)eps, (const T*)g_x_out, (const T*)a,
                           (T*)g_uv_bt);
      else
        hipLaunchKernelGGL((k_uv_reduce_bwd_c<T, NODE_NPB>), dim3((unsigned)((n + NODE_NPB - 1) / NODE_NPB)), dim3(256), 0,
                           (hipStream_t)stream, (const T*)uv_bt, (const T*)g_p, (const T*)g_cat, ld_cat, node_dim, n, ir,
                           (T)eps, (const T*)g_x_out, (const T*)a, (T*)g_uv_bt);
    } else
      hipLaunchKernelGGL((k_uv_reduce_bwd<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)uv_bt, (const T*)g_p, (const T*)g_cat, ld_cat, node_dim, n, ir, (T)eps,
                         (const T*)g_x_out, (const T*)a, (T*)g_uv_bt);
  });
  XEQ_CHECK_LAUNCH("xeq_uv_reduce_bwd");
  return XEQ_OK;
}

int xeq_update_out_fwd(int dtype, const void* s, const void* x, const void* uv_bt, const void* a, const void* ip,
                       int64_t n, int node_dim, const int32_t mul[3], void* s_out, void* x_out, void* stream) {
  XEQ_IR("xeq_update_out_fwd");
  if (n <= 0) return XEQ_OK;
  const int bs = 256;   // (2 x 320 for the 608 columns of the default model measured no better than 3 x 256)
  const int ncb = (node_dim + ir.D() + bs - 1) / bs;
  const int npb = node_npb(n);
  const int64_t nblk = (n + npb - 1) / npb * ncb;
  XEQ_CHECK_ARG(nblk < (1ll << 31), "xeq_update_out_fwd: too many nodes for one launch");
  XEQ_DISPATCH_FLOAT(dtype, {
    if (npb == 1)
      hipLaunchKernelGGL((k_update_out_fwd_c<T, 1>), dim3((unsigned)nblk), dim3(bs), 0, (hipStream_t)stream, (const T*)s,
                         (const T*)x, (const T*)uv_bt, (const T*)a, (const T*)ip, n, node_dim, ir, ncb, (T*)s_out, (T*)x_out);
    else
      hipLaunchKernelGGL((k_update_out_fwd_c<T, NODE_NPB>), dim3((unsigned)nblk), dim3(bs), 0, (hipStream_t)stream, (const T*)s,
                         (const T*)x, (const T*)uv_bt, (const T*)a, (const T*)ip, n, node_dim, ir, ncb, (T*)s_out, (T*)x_out);
  });
  XEQ_CHECK_LAUNCH("xeq_update_out_fwd");
  return XEQ_OK;
}

int xeq_update_out_bwd(int dtype, const void* g_s_out, const void* g_x_out, const void* uv_bt, const void* a,
                       const void* ip, int64_t n, int node_dim, const int32_t mul[3], void* g_a, void* g_ip,
                       void* g_uv_bt, void* stream) {
  XEQ_IR("xeq_update_out_bwd");
  if (n <= 0) return XEQ_OK;
  const int bs = node_block(node_dim + ir.C());
  const int ncb = (node_dim + ir.C() + bs - 1) / bs;
  const int npb = node_npb(n);
  const int64_t nblk = (n + npb - 1) / npb * ncb;
  XEQ_CHECK_ARG(nblk < (1ll << 31), "xeq_update_out_bwd: too many nodes for one launch");
  XEQ_DISPATCH_FLOAT(dtype, {
    if (npb == 1)
      hipLaunchKernelGGL((k_update_out_bwd_c<T, 1>), dim3((unsigned)nblk), dim3(bs), 0, (hipStream_t)stream,
                         (const T*)g_s_out, (const T*)g_x_out, (const T*)uv_bt, (const T*)a, (const T*)ip, n, node_dim, ir,
                         ncb, (T*)g_a, (T*)g_ip, (T*)g_uv_bt);
    else
      hipLaunchKernelGGL((k_update_out_bwd_c<T, NODE_NPB>), dim3((unsigned)nblk), dim3(bs), 0, (hipStream_t)stream,
                         (const T*)g_s_out, (const T*)g_x_out, (const T*)uv_bt, (const T*)a, (const T*)ip, n, node_dim, ir,
                         ncb, (T*)g_a, (T*)g_ip, (T*)g_uv_bt);
  });
  XEQ_CHECK_LAUNCH("xeq_update_out_bwd");
  return XEQ_OK;
}

}  // extern "C"
