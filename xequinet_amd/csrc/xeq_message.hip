// Fused XPaiNN message kernels, GENERIC form (SURVEY 8a rows a3-a5, a10-a13, and their reverse
// pass for a16).  Reference dataflow: nn/xpainn.py:140-159.
//
// This is the fallback for configurations neither xeq_message_wq.hip (f32, multiplicities in multiples of 32) nor
// xeq_message_sb.hip (at most 256 channels) covers: any multiplicities, any node_dim, f32 and f64.
//
// Mapping ("channel on the lane"):
//   * one 256-thread workgroup walks destination nodes (persistent grid,
//     XCD-aware node->workgroup map so a molecule's rows stay in one L2);
//   * thread t owns gate channel u = t (its gate_state and gate_edge filter rows
//     and the 2l+1 components of x[., u, .]) and scalar channel t of msg_s; the
//     three rbf_lin rows live in registers for the whole launch;
//   * per-edge quantities that every channel needs (f*rho_k(d), Y_lm(r), and in
//     the reverse pass their derivatives) are computed once per edge by spare
//     lanes and broadcast through LDS -- rbf[E,20], fcut[E], rsh[E,480] never
//     exist in HBM;
//   * the per-destination reduction is a register accumulation over the node's
//     CSR segment: no atomics, bitwise reproducible.
#include "xeq_common.h"

namespace xeq {

constexpr int EC = 32;    // edges staged per chunk
constexpr int YS = 12;    // stride of the per-edge SH record: [1 | Y1(3) | Y2(5) | f | f' | pad]

struct MsgArgs {
  int64_t n_nodes, n_edges;
  const int32_t* rowptr;
  const int32_t* perm;
  const int64_t* other;  // fwd: neighbor index per edge; bwd: center index per edge
  int F, C, D, H;
  Irreps ir;
  RadialSpec rs;
  int chunk;  // consecutive nodes kept on one XCD label
  int xl;     // layout of xhat / grad_xhat (see XAddr)
};

template <typename T, int MAXB>
__device__ __forceinline__ void load_w_row(const T* __restrict__ w, const T* __restrict__ b, int row, int B,
                                           bool valid, T (&wr)[MAXB], T& br) {
#pragma unroll
  for (int k = 0; k < MAXB; ++k) wr[k] = (valid && k < B) ? w[(int64_t)row * B + k] : T(0);
  br = valid ? b[row] : T(0);
}

// stage one chunk of edges into LDS: geometry + SH record, then (edge, k) radial terms
template <typename T, int MAXB, bool BWD>
__device__ __forceinline__ void stage_chunk(const MsgArgs& a, const T* __restrict__ vec, const T* __restrict__ p0,
                                            const T* __restrict__ p1, int32_t base, int cnt, T (*sh_rf)[MAXB],
                                            T (*sh_drf)[MAXB], T (*sh_y)[YS], T (*sh_g)[5], int32_t* sh_other,
                                            int32_t* sh_eid) {
  const int t = threadIdx.x;
  const T rc = (T)a.rs.cutoff;
  if (t < cnt) {
    int32_t e = a.perm ? a.perm[base + t] : base + t;
    EdgeGeom<T> g = edge_geom<T>(vec[3 * (int64_t)e], vec[3 * (int64_t)e + 1], vec[3 * (int64_t)e + 2]);
    T y1[3], y2[5], f, df;
    sph_harm_l12<T>(g, y1, y2);
    envelope<T>(a.rs.cutoff_kind, g.d, rc, f, df);
    sh_y[t][0] = T(1);
#pragma unroll
    for (int m = 0; m < 3; ++m) sh_y[t][1 + m] = y1[m];
#pragma unroll
    for (int m = 0; m < 5; ++m) sh_y[t][4 + m] = y2[m];
    sh_y[t][9] = f;
    sh_y[t][10] = df;
    sh_g[t][0] = g.x;
    sh_g[t][1] = g.y;
    sh_g[t][2] = g.z;
    sh_g[t][3] = g.d;
    sh_g[t][4] = g.inv_d;
    sh_other[t] = (int32_t)a.other[e];
    sh_eid[t] = e;
  }
  __syncthreads();
  const int B = a.rs.num_basis;
  for (int idx = t; idx < cnt * B; idx += blockDim.x) {
    int j = idx / B, k = idx - j * B;
    T d = sh_g[j][3], f = sh_y[j][9], df = sh_y[j][10];
    T rho, drho;
    radial<T>(a.rs.rbf_kind, d, rc, p0[k], p1 ? p1[k] : T(0), rho, drho, k, B);
    sh_rf[j][k] = f * rho;
    if (BWD) sh_drf[j][k] = df * rho + f * drho;
  }
  __syncthreads();
}

template <typename T, int MAXB>
__global__ void __launch_bounds__(256) k_message_fwd(MsgArgs a, const T* __restrict__ vec, const T* __restrict__ h,
                                                     const T* __restrict__ xhat, const T* __restrict__ s_in,
                                                     const T* __restrict__ x_in, const T* __restrict__ w_rbf,
                                                     const T* __restrict__ b_rbf, const T* __restrict__ p0,
                                                     const T* __restrict__ p1, T* __restrict__ s_out,
                                                     T* __restrict__ x_out) {
  __shared__ T sh_rf[EC][MAXB];
  __shared__ T sh_y[EC][YS];
  __shared__ T sh_g[EC][5];
  __shared__ int32_t sh_nbr[EC];
  __shared__ int32_t sh_eid[EC];
  const int t = threadIdx.x;
  const int B = a.rs.num_basis, C = a.C, F = a.F, D = a.D, H = a.H;
  const bool has_u = t < C, has_s = t < F;
  int l = 0, off = 0;
  if (has_u) a.ir.locate(t, l, off);
  const int nm = has_u ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const XAddr xa = xaddr(a.ir, a.n_nodes, has_u ? t : 0, a.xl);

  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w_row<T, MAXB>(w_rbf, b_rbf, t, B, has_u, ws, bs);
  load_w_row<T, MAXB>(w_rbf, b_rbf, C + t, B, has_u, we, be);
  load_w_row<T, MAXB>(w_rbf, b_rbf, 2 * C + t, B, has_s, wm, bm);
  // zero the padded radial columns once (never written afterwards)
  for (int idx = t; idx < EC * MAXB; idx += blockDim.x) sh_rf[idx / MAXB][idx % MAXB] = T(0);
  __syncthreads();

  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t c = walk.next(); c >= 0; c = walk.next()) {
    const int32_t e0 = a.rowptr[c], e1 = a.rowptr[c + 1];
    T acc_s = T(0);
    T acc_x[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t base = e0; base < e1; base += EC) {
      const int cnt = min(EC, e1 - base);
      __syncthreads();  // previous chunk fully consumed
      stage_chunk<T, MAXB, false>(a, vec, p0, p1, base, cnt, sh_rf, nullptr, sh_y, sh_g, sh_nbr, sh_eid);
      for (int j = 0; j < cnt; ++j) {
        const int64_t n = sh_nbr[j];
        const T f = sh_y[j][9];
        T ds = bs * f, de = be * f, dm = bm * f;
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
          T r = sh_rf[j][k];
          ds += ws[k] * r;
          de += we[k] * r;
          dm += wm[k] * r;
        }
        const T* hn = h + n * H;
        if (has_s) acc_s += hn[2 * C + t] * dm;
        if (has_u) {
          T gs = hn[t] * ds, ge = hn[C + t] * de;
          const T* xn = xhat + xa.off + n * xa.node;
#pragma unroll
          for (int m = 0; m < 5; ++m)
            if (m < nm) acc_x[m] += xn[m * xa.comp] * gs + sh_y[j][yoff + m] * ge;
        }
      }
    }
    if (has_s) s_out[c * F + t] = s_in[c * F + t] + acc_s;
    if (has_u) {
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < nm) x_out[c * D + off + m] = x_in[c * D + off + m] + acc_x[m];
    }
  }
}

template <typename T, int MAXB>
__global__ void __launch_bounds__(256) k_message_bwd(MsgArgs a, const T* __restrict__ vec, const T* __restrict__ h,
                                                     const T* __restrict__ xhat, const T* __restrict__ grad_s,
                                                     const T* __restrict__ grad_x, const T* __restrict__ w_rbf,
                                                     const T* __restrict__ b_rbf, const T* __restrict__ p0,
                                                     const T* __restrict__ p1, T* __restrict__ grad_h,
                                                     T* __restrict__ grad_xhat, T* __restrict__ grad_vec) {
  __shared__ T sh_rf[EC][MAXB];
  __shared__ T sh_drf[EC][MAXB];
  __shared__ T sh_y[EC][YS];
  __shared__ T sh_g[EC][5];
  __shared__ int32_t sh_ctr[EC];
  __shared__ int32_t sh_eid[EC];
  __shared__ T sh_red[EC][4][9];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.rs.num_basis, C = a.C, F = a.F, D = a.D, H = a.H;
  const bool has_u = t < C, has_s = t < F;
  int l = 0, off = 0;
  if (has_u) a.ir.locate(t, l, off);
  const int nm = has_u ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const bool wave_has1 = __ballot(has_u && l == 1) != 0ull;
  const bool wave_has2 = __ballot(has_u && l == 2) != 0ull;
  const XAddr xa = xaddr(a.ir, a.n_nodes, has_u ? t : 0, a.xl);

  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w_row<T, MAXB>(w_rbf, b_rbf, t, B, has_u, ws, bs);
  load_w_row<T, MAXB>(w_rbf, b_rbf, C + t, B, has_u, we, be);
  load_w_row<T, MAXB>(w_rbf, b_rbf, 2 * C + t, B, has_s, wm, bm);
  for (int idx = t; idx < EC * MAXB; idx += blockDim.x) {
    sh_rf[idx / MAXB][idx % MAXB] = T(0);
    sh_drf[idx / MAXB][idx % MAXB] = T(0);
  }
  __syncthreads();

  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t n = walk.next(); n >= 0; n = walk.next()) {
    const int32_t e0 = a.rowptr[n], e1 = a.rowptr[n + 1];
    // this node's own rows (it is the NEIGHBOR/source of every edge in the segment)
    const T hs = has_u ? h[n * H + t] : T(0), he = has_u ? h[n * H + C + t] : T(0);
    const T hm = has_s ? h[n * H + 2 * C + t] : T(0);
    T xh[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) xh[m] = (m < nm) ? xhat[xa.off + n * xa.node + m * xa.comp] : T(0);
    T acc_hs = T(0), acc_he = T(0), acc_hm = T(0);
    T acc_xh[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t base = e0; base < e1; base += EC) {
      const int cnt = min(EC, e1 - base);
      __syncthreads();
      stage_chunk<T, MAXB, true>(a, vec, p0, p1, base, cnt, sh_rf, sh_drf, sh_y, sh_g, sh_ctr, sh_eid);
      for (int j = 0; j < cnt; ++j) {
        const int64_t c = sh_ctr[j];
        const T f = sh_y[j][9], df = sh_y[j][10];
        T ps = bs * f, pe = be * f, pm = bm * f;        // filter phi
        T qs = bs * df, qe = be * df, qm = bm * df;     // d phi / d d
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
          T r = sh_rf[j][k], dr = sh_drf[j][k];
          ps += ws[k] * r;
          pe += we[k] * r;
          pm += wm[k] * r;
          qs += ws[k] * dr;
          qe += we[k] * dr;
          qm += wm[k] * dr;
        }
        T gx[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) gx[m] = (m < nm) ? grad_x[c * D + off + m] : T(0);
        const T dgm = has_s ? grad_s[c * F + t] : T(0);
        T dgs = T(0), dge = T(0);
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          T yv = (m < nm) ? sh_y[j][yoff + m] : T(0);
          dgs += xh[m] * gx[m];
          dge += yv * gx[m];
        }
        acc_hs += ps * dgs;
        acc_he += pe * dge;
        acc_hm += pm * dgm;
        const T gate = hs * ps;
#pragma unroll
        for (int m = 0; m < 5; ++m) acc_xh[m] += gate * gx[m];
        // per-edge scalars: dL/dd and dL/dY_lm, reduced over channels
        T pd = hs * dgs * qs + he * dge * qe + hm * dgm * qm;
        pd = wave_sum<T>(pd);
        const T gy = he * pe;
        T r1[3] = {T(0), T(0), T(0)}, r2[5] = {T(0), T(0), T(0), T(0), T(0)};
        if (wave_has1) {
#pragma unroll
          for (int m = 0; m < 3; ++m) r1[m] = wave_sum<T>(l == 1 ? gy * gx[m] : T(0));
        }
        if (wave_has2) {
#pragma unroll
          for (int m = 0; m < 5; ++m) r2[m] = wave_sum<T>(l == 2 ? gy * gx[m] : T(0));
        }
        if (lane == 0) {
          sh_red[j][wave][0] = pd;
#pragma unroll
          for (int m = 0; m < 3; ++m) sh_red[j][wave][1 + m] = r1[m];
#pragma unroll
          for (int m = 0; m < 5; ++m) sh_red[j][wave][4 + m] = r2[m];
        }
      }
      __syncthreads();
      if (t < cnt) {
        T gd = T(0), q1[3] = {T(0), T(0), T(0)}, q2[5] = {T(0), T(0), T(0), T(0), T(0)};
        for (int w = 0; w < 4; ++w) {
          gd += sh_red[t][w][0];
#pragma unroll
          for (int m = 0; m < 3; ++m) q1[m] += sh_red[t][w][1 + m];
#pragma unroll
          for (int m = 0; m < 5; ++m) q2[m] += sh_red[t][w][4 + m];
        }
        EdgeGeom<T> g;
        g.x = sh_g[t][0];
        g.y = sh_g[t][1];
        g.z = sh_g[t][2];
        g.d = sh_g[t][3];
        g.inv_d = sh_g[t][4];
        T out[3];
        edge_grad<T>(g, gd, q1, q2, out);
        const int64_t e = sh_eid[t];
        grad_vec[3 * e] = out[0];
        grad_vec[3 * e + 1] = out[1];
        grad_vec[3 * e + 2] = out[2];
      }
    }
    if (has_u) {
      grad_h[n * H + t] = acc_hs;
      grad_h[n * H + C + t] = acc_he;
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < nm) grad_xhat[xa.off + n * xa.node + m * xa.comp] = acc_xh[m];
    }
    if (has_s) grad_h[n * H + 2 * C + t] = acc_hm;
  }
}

// ---- parameter gradients of the radial filter (training pass) -------------------------------------------------------
// filter[e, c] = (sum_k W[c, k] rho_k(d_e) + b[c]) f(d_e)   (nn/xpainn.py:140), filter_out = h[nbr] * filter.  With
//   G[e, c] = dL/dfilter_out[e, c] h[nbr(e), c]:   dL/dfilter_out = <grad_x[ctr], xhat[nbr]>_m (gate_state rows), <grad_x[ctr], Y(e)>_m
//   (gate_edge rows), grad_s[ctr] (msg_s rows) -- the same products the reverse kernel forms for dL/dh,
// the launch writes per workgroup and filter row c
//   parts[p][c][0..B)       sum_e G f rho_k              -> dL/dW[c, k]
//   parts[p][c][B]          sum_e G f                    -> dL/db[c]
//   parts[p][c][B+1..2B+1)  sum_e G f d rho_k / d p0_k   -> dL/dp0_k = sum_c W[c, k] (.)      (freq / mean)
//   parts[p][c][2B+1..3B+1) sum_e G f d rho_k / d p1_k   -> dL/dp1_k likewise                 (std; zero for the Bessel basis)
// and the caller adds the parts up (a fixed order: bitwise reproducible).  This is the only parameter gradient of the model that is
// not a contraction over saved NODE tensors (the [E, 576] operand never exists).  Mapping as k_message_bwd: a workgroup walks source
// nodes, a thread owns one filter row of its role (blockIdx.y: 0 gate_state, 1 gate_edge, 2 msg_s) with 3 B + 1 running sums in
// registers; the per-edge factors are computed once per chunk by spare lanes and broadcast through LDS.
template <typename T, int MAXB, bool HAS_P1>
__global__ void __launch_bounds__(256) k_message_param_grad(MsgArgs a, const T* __restrict__ vec, const T* __restrict__ h,
                                                            const T* __restrict__ xhat, const T* __restrict__ grad_s,
                                                            const T* __restrict__ grad_x, const T* __restrict__ p0,
                                                            const T* __restrict__ p1, T* __restrict__ parts) {
  __shared__ T sh_rf[EC][MAXB];
  __shared__ T sh_q0[EC][MAXB];
  __shared__ T sh_q1[HAS_P1 ? EC : 1][MAXB];
  __shared__ T sh_y[EC][YS];
  __shared__ T sh_d[EC];
  __shared__ int32_t sh_ctr[EC];
  const int t = threadIdx.x, role = blockIdx.y;
  const int B = a.rs.num_basis, C = a.C, F = a.F, D = a.D, H = a.H;
  const bool active = role == 2 ? t < F : t < C;
  const int row = role * C + t;   // filter row: [gate_state C | gate_edge C | msg_s F]
  int l = 0, off = 0;
  if (role < 2 && active) a.ir.locate(t, l, off);
  const int nm = (role < 2 && active) ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const XAddr xa = xaddr(a.ir, a.n_nodes, (role < 2 && active) ? t : 0, a.xl);
  const T rc = (T)a.rs.cutoff;
  T aw[MAXB], a0[MAXB], a1[HAS_P1 ? MAXB : 1], ab = T(0);
#pragma unroll
  for (int k = 0; k < MAXB; ++k) aw[k] = a0[k] = T(0);
#pragma unroll
  for (int k = 0; k < (HAS_P1 ? MAXB : 1); ++k) a1[k] = T(0);
  for (int idx = t; idx < EC * MAXB; idx += blockDim.x) {
    sh_rf[idx / MAXB][idx % MAXB] = T(0);
    sh_q0[idx / MAXB][idx % MAXB] = T(0);
    if (HAS_P1) sh_q1[idx / MAXB][idx % MAXB] = T(0);
  }
  __syncthreads();

  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t n = walk.next(); n >= 0; n = walk.next()) {
    const int32_t e0 = a.rowptr[n], e1 = a.rowptr[n + 1];
    const T hrow = active ? h[n * H + row] : T(0);
    T xh[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) xh[m] = (role == 0 && m < nm) ? xhat[xa.off + n * xa.node + m * xa.comp] : T(0);
    for (int32_t base = e0; base < e1; base += EC) {
      const int cnt = min(EC, e1 - base);
      __syncthreads();   // previous chunk fully consumed
      if (t < cnt) {
        const int32_t e = a.perm ? a.perm[base + t] : base + t;
        EdgeGeom<T> g = edge_geom<T>(vec[3 * (int64_t)e], vec[3 * (int64_t)e + 1], vec[3 * (int64_t)e + 2]);
        T y1[3], y2[5], f, df;
        sph_harm_l12<T>(g, y1, y2);
        envelope<T>(a.rs.cutoff_kind, g.d, rc, f, df);
        sh_y[t][0] = T(1);
#pragma unroll
        for (int m = 0; m < 3; ++m) sh_y[t][1 + m] = y1[m];
#pragma unroll
        for (int m = 0; m < 5; ++m) sh_y[t][4 + m] = y2[m];
        sh_y[t][9] = f;
        sh_d[t] = g.d;
        sh_ctr[t] = (int32_t)a.other[e];
      }
      __syncthreads();
      for (int idx = t; idx < cnt * B; idx += blockDim.x) {
        const int j = idx / B, k = idx - j * B;
        const T d = sh_d[j], f = sh_y[j][9];
        const T q0 = p0[k], q1 = p1 ? p1[k] : T(0);
        T rho, drho, d0, d1;
        radial<T>(a.rs.rbf_kind, d, rc, q0, q1, rho, drho);
        radial_dparam<T>(a.rs.rbf_kind, d, rc, q0, q1, d0, d1);
        sh_rf[j][k] = f * rho;
        sh_q0[j][k] = f * d0;
        if (HAS_P1) sh_q1[j][k] = f * d1;
      }
      __syncthreads();
      for (int j = 0; j < cnt; ++j) {
        const int64_t c = sh_ctr[j];
        T dg;
        if (role == 2) {
          dg = active ? grad_s[c * F + t] : T(0);
        } else {
          dg = T(0);
#pragma unroll
          for (int m = 0; m < 5; ++m) {
            if (m < nm) {
              const T gx = grad_x[c * D + off + m];
              dg += (role == 0 ? xh[m] : sh_y[j][yoff + m]) * gx;
            }
          }
        }
        const T G = hrow * dg;
        ab += G * sh_y[j][9];
#pragma unroll
        for (int k = 0; k < MAXB; ++k) {
          aw[k] += G * sh_rf[j][k];
          a0[k] += G * sh_q0[j][k];
        }
        if (HAS_P1) {
#pragma unroll
          for (int k = 0; k < MAXB; ++k) a1[k] += G * sh_q1[j][k];
        }
      }
    }
  }
  if (active) {
    T* out = parts + ((int64_t)blockIdx.x * H + row) * (3 * B + 1);
#pragma unroll
    for (int k = 0; k < MAXB; ++k)
      if (k < B) {
        out[k] = aw[k];
        out[B + 1 + k] = a0[k];
        out[2 * B + 1 + k] = HAS_P1 ? a1[HAS_P1 ? k : 0] : T(0);
      }
    out[B] = ab;
  }
}

static int check_msg(const char* who, int64_t n_nodes, int64_t n_edges, int num_basis, double cutoff,
                     int rbf_kind, int cutoff_kind, int node_dim, const int32_t mul[3], const void* p1,
                     MsgArgs& a) {
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (1ll << 31) && n_nodes < (1ll << 31), "%s: bad sizes", who);
  XEQ_CHECK_ARG(num_basis >= 1 && num_basis <= 32, "%s: num_basis %d outside the supported range 1..32", who, num_basis);
  XEQ_CHECK_ARG(cutoff > 0, "%s: cutoff must be positive", who);
  XEQ_CHECK_ARG(rbf_kind >= XEQ_RBF_BESSEL && rbf_kind <= XEQ_RBF_EXPNORM, "%s: rbf kernel %d is not implemented", who, rbf_kind);
  XEQ_CHECK_ARG(rbf_kind == XEQ_RBF_BESSEL || p1 != nullptr, "%s: this radial basis needs its second parameter array (std / logc / mu)", who);
  XEQ_CHECK_ARG(cutoff_kind == XEQ_CUTOFF_COSINE || cutoff_kind == XEQ_CUTOFF_POLYNOMIAL, "%s: cutoff function %d is not implemented", who, cutoff_kind);
  for (int l = 0; l < 3; ++l) {
    XEQ_CHECK_ARG(mul[l] >= 0, "%s: negative multiplicity", who);
    a.ir.mul[l] = mul[l];
  }
  a.C = a.ir.C();
  a.D = a.ir.D();
  a.F = node_dim;
  a.H = a.F + 2 * a.C;
  XEQ_CHECK_ARG(a.C >= 1 && a.C <= 256 && a.F >= 1 && a.F <= 256,
                "%s: node_dim %d / %d irrep channels exceed the 256-channel workgroup mapping", who, a.F, a.C);
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.rs = RadialSpec{rbf_kind, cutoff_kind, num_basis, cutoff};
  a.chunk = n_nodes >= 32 * 1024 ? 32 : (int)(n_nodes / 1024 > 0 ? n_nodes / 1024 : 1);
  return XEQ_OK;
}

static inline unsigned msg_grid(int64_t n_nodes) {
  int64_t g = 256 * 4;
  return (unsigned)(n_nodes < g ? n_nodes : g);
}

}  // namespace xeq

using namespace xeq;

#define XEQ_MSG_DISPATCH_B(KERNEL, ...)                                                                   \
  do {                                                                                                    \
    if (num_basis <= 8) hipLaunchKernelGGL((KERNEL<T, 8>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);   \
    else if (num_basis <= 16) hipLaunchKernelGGL((KERNEL<T, 16>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (num_basis <= 20) hipLaunchKernelGGL((KERNEL<T, 20>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<T, 32>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
  } while (0)

extern "C" {

int xeq_message_fwd(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                    const int64_t* nbr, const void* vec, const void* h, const void* xhat, const void* s_in,
                    const void* x_in, const void* w_rbf, const void* b_rbf, const void* p0, const void* p1,
                    int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                    const int32_t mul[3], void* s_out, void* x_out, int xhat_layout, void* stream) {
  MsgArgs a{};
  int rcode = check_msg("xeq_message_fwd", n_nodes, n_edges, num_basis, cutoff, rbf_kind, cutoff_kind, node_dim,
                        mul, p1, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = rowptr;
  a.perm = perm;
  a.other = nbr;
  a.xl = xhat_layout & 1;   // the XEQ_XHAT_HIGHER_L_ZERO hint is for the wq kernels; this family computes the general form
  dim3 grid(msg_grid(n_nodes));
  XEQ_DISPATCH_FLOAT(dtype, {
    XEQ_MSG_DISPATCH_B(k_message_fwd, a, (const T*)vec, (const T*)h, (const T*)xhat, (const T*)s_in,
                       (const T*)x_in, (const T*)w_rbf, (const T*)b_rbf, (const T*)p0, (const T*)p1, (T*)s_out,
                       (T*)x_out);
  });
  XEQ_CHECK_LAUNCH("xeq_message_fwd");
  return XEQ_OK;
}

int xeq_message_bwd(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                    const int64_t* center, const void* vec, const void* h, const void* xhat, const void* grad_s,
                    const void* grad_x, const void* w_rbf, const void* b_rbf, const void* p0, const void* p1,
                    int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                    const int32_t mul[3], void* grad_h, void* grad_xhat, void* grad_vec, int xhat_layout, void* stream) {
  MsgArgs a{};
  int rcode = check_msg("xeq_message_bwd", n_nodes, n_edges, num_basis, cutoff, rbf_kind, cutoff_kind, node_dim,
                        mul, p1, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = n_rowptr;
  a.perm = n_perm;
  a.other = center;
  a.xl = xhat_layout & 1;   // the XEQ_XHAT_HIGHER_L_ZERO hint is for the wq kernels; this family computes the general form
  dim3 grid(msg_grid(n_nodes));
  XEQ_DISPATCH_FLOAT(dtype, {
    XEQ_MSG_DISPATCH_B(k_message_bwd, a, (const T*)vec, (const T*)h, (const T*)xhat, (const T*)grad_s,
                       (const T*)grad_x, (const T*)w_rbf, (const T*)b_rbf, (const T*)p0, (const T*)p1,
                       (T*)grad_h, (T*)grad_xhat, (T*)grad_vec);
  });
  XEQ_CHECK_LAUNCH("xeq_message_bwd");
  return XEQ_OK;
}


int xeq_message_param_grad_parts(int64_t n_nodes) {
  const int64_t g = 256 * 2;
  return (int)(n_nodes < g ? (n_nodes > 0 ? n_nodes : 1) : g);
}

int xeq_message_param_grad(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                           const int64_t* center, const void* vec, const void* h, const void* xhat, const void* grad_s,
                           const void* grad_x, const void* p0, const void* p1, int rbf_kind, int cutoff_kind, int num_basis,
                           double cutoff, int node_dim, const int32_t mul[3], int xhat_layout, int n_parts, void* parts,
                           void* stream) {
  MsgArgs a{};
  int rcode = check_msg("xeq_message_param_grad", n_nodes, n_edges, num_basis, cutoff, rbf_kind, cutoff_kind, node_dim, mul, p1, a);
  XEQ_CHECK_ARG(rbf_kind == XEQ_RBF_BESSEL || rbf_kind == XEQ_RBF_GAUSSIAN, "xeq_message_param_grad: the parameter gradients of rbf kernel %d are not built (the differentiable tensor form takes that basis)", rbf_kind);
  if (rcode != XEQ_OK) return rcode;
  XEQ_CHECK_ARG(n_parts == xeq_message_param_grad_parts(n_nodes), "xeq_message_param_grad: parts must hold xeq_message_param_grad_parts(n_nodes) = %d blocks, got %d",
                xeq_message_param_grad_parts(n_nodes), n_parts);
  a.rowptr = n_rowptr;
  a.perm = n_perm;
  a.other = center;
  a.xl = xhat_layout & 1;
  const size_t bytes = (size_t)n_parts * a.H * (3 * num_basis + 1) * (dtype == XEQ_F64 ? 8 : 4);
  if (n_nodes == 0) {   // nothing to walk: the sums are zero
    if (hipMemsetAsync(parts, 0, bytes, (hipStream_t)stream) != hipSuccess) return XEQ_ERR_LAUNCH;
    return XEQ_OK;
  }
  dim3 grid((unsigned)n_parts, 3);
#define XEQ_PG_LAUNCH(MAXB, P1)                                                                                                   \
  hipLaunchKernelGGL((k_message_param_grad<T, MAXB, P1>), grid, dim3(256), 0, (hipStream_t)stream, a, (const T*)vec, (const T*)h, \
                     (const T*)xhat, (const T*)grad_s, (const T*)grad_x, (const T*)p0, (const T*)p1, (T*)parts)
#define XEQ_PG_DISPATCH(P1)                                \
  do {                                                     \
    if (num_basis <= 8) XEQ_PG_LAUNCH(8, P1);              \
    else if (num_basis <= 16) XEQ_PG_LAUNCH(16, P1);       \
    else if (num_basis <= 20) XEQ_PG_LAUNCH(20, P1);       \
    else XEQ_PG_LAUNCH(32, P1);                            \
  } while (0)
  XEQ_DISPATCH_FLOAT(dtype, {
    if (rbf_kind == XEQ_RBF_GAUSSIAN) XEQ_PG_DISPATCH(true);
    else XEQ_PG_DISPATCH(false);
  });
  XEQ_CHECK_LAUNCH("xeq_message_param_grad");
  return XEQ_OK;
}

}  // extern "C"
