// Fused XPaiNN message kernels, fp32 MFMA edition (default path for float32).
// Reference dataflow: nn/xpainn.py:140-159; reverse pass for nn/basic.py:143-159.
//
// Per 16-edge tile of a destination-sorted (forward) / source-sorted (reverse) edge
// list, a 256-thread workgroup runs three phases:
//   0  every (edge, k) pair gets one lane: |r|, envelope, radial term f*rho_k (and
//      its d/dd in the reverse pass) -> LDS; the k = 0 lane also stores Y_lm, the
//      neighbour / centre index and the unit vector;
//   1  the rbf_lin contraction filter[c, e] = sum_k W[c, k] f rho_k(e) + b[c] f(e) on
//      the matrix cores: v_mfma_f32_16x16x4_f32, A = 16 channels x 4 k (register
//      resident for the whole launch), B = 4 k x 16 edges, C initialised with the
//      bias term; each wave owns 9 of the 36 channel tiles; D is written to LDS as
//      filter[e][c] (row stride = 4 mod 32 dwords: conflict-free b128 stores);
//   2  "channel on the lane": thread t owns gate channel t and scalar channel t,
//      walks the tile's edges, gathers h[nbr] / xhat[nbr] rows (software prefetch
//      one edge ahead), gates, and accumulates the node's segment in registers.
// A workgroup owns a contiguous node range chosen so that every workgroup gets the
// same number of edges (binary search on rowptr): no tail, no atomics, results are
// bitwise reproducible.  rbf[E,20], fcut[E], rsh[E,480], filter[E,576] never touch HBM.
#include "xeq_message_tile.h"

namespace xeq {

template <int KS>
__global__ void __launch_bounds__(256) k_message_fwd_mfma(Msg2Args a, const float* __restrict__ vec,
                                                          const float* __restrict__ h, const float* __restrict__ xhat,
                                                          const float* __restrict__ s_in, const float* __restrict__ x_in,
                                                          const float* __restrict__ w_rbf, const float* __restrict__ b_rbf,
                                                          const float* __restrict__ p0, const float* __restrict__ p1,
                                                          float* __restrict__ s_out, float* __restrict__ x_out) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ Smem<KS, false> sm;
  float* sh_phi = dyn;                 // [TE][HP]
  float* sh_bias = dyn + TE * a.HP;    // [NT*16]
  const int t = threadIdx.x;
  const int C = a.C, F = a.F, D = a.D, H = a.H, HP = a.HP;
  const bool has_u = t < C, has_s = t < F;
  int l = 0, off = 0;
  if (has_u) a.ir.locate(t, l, off);
  const int nm = has_u ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const XAddr xa = xaddr(a.ir, a.n_nodes, has_u ? t : 0, a.xl);

  float wa[TPW][KS];
  load_a_frags<KS>(a, w_rbf, wa);
  for (int c = t; c < a.NT * 16; c += 256) sh_bias[c] = c < a.H ? b_rbf[c] : 0.f;
  // radial constants of this thread's basis index (phase 0): k = t % (4 KS)
  const int kq = t % (4 * KS);
  const float p0k = kq < a.rs.num_basis ? p0[kq] : 0.f;
  const float p1k = (p1 && kq < a.rs.num_basis) ? p1[kq] : 0.f;
  const float wk = (float)((double)p0k * 0.15915494309189535);  // p0 / (2 pi)
  if (t < 2) {  // equal-edge split of the node range over the grid
    int64_t target = (a.n_edges * (int64_t)(blockIdx.x + t) + gridDim.x - 1) / gridDim.x;
    int32_t n = (blockIdx.x + t == 0) ? 0 : ((blockIdx.x + t == gridDim.x) ? (int32_t)a.n_nodes : lower_node(a.rowptr, a.n_nodes, target));
    sm.range[t] = n;
  }
  __syncthreads();
  const int64_t n0 = sm.range[0], n1 = sm.range[1];
  if (n0 >= n1) return;  // uniform
  const int32_t e_begin = a.rowptr[n0], e_end = a.rowptr[n1];

  int64_t cur = n0;
  float acc_s = 0.f, acc_x[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  auto flush = [&](int64_t c) {
    if (has_s) s_out[c * F + t] = s_in[c * F + t] + acc_s;
    if (has_u) {
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < nm) x_out[c * D + off + m] = x_in[c * D + off + m] + acc_x[m];
    }
    acc_s = 0.f;
#pragma unroll
    for (int m = 0; m < 5; ++m) acc_x[m] = 0.f;
  };

  for (int32_t base = e_begin; base < e_end; base += TE) {
    const int cnt = min(TE, e_end - base);
    __syncthreads();  // previous tile fully consumed
    phase0<KS, false>(a, vec, p0, p1, base, cnt, n0, n1, sm, wk, p0k, p1k);
    __syncthreads();
    if (!(a.ablate & 2)) phase1<KS, false>(a, wa, sh_bias, sm, sh_phi, nullptr);
    __syncthreads();
    if (a.ablate & 4) continue;
    // ---- phase 2: batches of GB edges; every gather of a batch is issued before its first use
    constexpr int GB = 4;
    for (int j0 = 0; j0 < cnt; j0 += GB) {
      float hs[GB], he[GB], hm[GB], xv[GB][5];
#pragma unroll
      for (int b = 0; b < GB; ++b) {
        const int j = min(j0 + b, cnt - 1);  // clamp: padded slots re-read a valid row and are skipped below
        const int64_t n = sm.other[j];
        const float* hn = h + n * H;
        hm[b] = has_s ? hn[2 * C + t] : 0.f;
        hs[b] = has_u ? hn[t] : 0.f;
        he[b] = has_u ? hn[C + t] : 0.f;
        const float* xn = xhat + xa.off + n * xa.node;
#pragma unroll
        for (int m = 0; m < 5; ++m) xv[b][m] = (m < nm) ? xn[m * xa.comp] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < GB; ++b) {
        const int j = j0 + b;
        if (j < cnt) {  // uniform
          const int64_t dst = sm.self[j];
          while (cur < dst) {  // uniform: segment boundary (also skips nodes without edges)
            flush(cur);
            ++cur;
          }
          const float* ph = sh_phi + j * HP;
          if (has_s) acc_s += hm[b] * ph[2 * C + t];
          if (has_u) {
            const float gs = hs[b] * ph[t], ge = he[b] * ph[C + t];
#pragma unroll
            for (int m = 0; m < 5; ++m)
              if (m < nm) acc_x[m] += xv[b][m] * gs + sm.y[j][yoff + m] * ge;
          }
        }
      }
    }
  }
  while (cur < n1) {
    flush(cur);
    ++cur;
  }
}

template <int KS>
__global__ void __launch_bounds__(256) k_message_bwd_mfma(Msg2Args a, const float* __restrict__ vec,
                                                          const float* __restrict__ h, const float* __restrict__ xhat,
                                                          const float* __restrict__ grad_s, const float* __restrict__ grad_x,
                                                          const float* __restrict__ w_rbf, const float* __restrict__ b_rbf,
                                                          const float* __restrict__ p0, const float* __restrict__ p1,
                                                          float* __restrict__ grad_h, float* __restrict__ grad_xhat,
                                                          float* __restrict__ grad_vec) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ Smem<KS, true> sm;
  float* sh_phi = dyn;                   // [TE][HP]
  float* sh_dphi = dyn + TE * a.HP;      // [TE][HP]
  float* sh_bias = dyn + 2 * TE * a.HP;  // [NT*16]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int C = a.C, F = a.F, D = a.D, H = a.H, HP = a.HP;
  const bool has_u = t < C, has_s = t < F;
  int l = 0, off = 0;
  if (has_u) a.ir.locate(t, l, off);
  const int nm = has_u ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const bool wave_has1 = __ballot(has_u && l == 1) != 0ull;
  const bool wave_has2 = __ballot(has_u && l == 2) != 0ull;
  const XAddr xa = xaddr(a.ir, a.n_nodes, has_u ? t : 0, a.xl);

  float wa[TPW][KS];
  load_a_frags<KS>(a, w_rbf, wa);
  for (int c = t; c < a.NT * 16; c += 256) sh_bias[c] = c < a.H ? b_rbf[c] : 0.f;
  // radial constants of this thread's basis index (phase 0): k = t % (4 KS)
  const int kq = t % (4 * KS);
  const float p0k = kq < a.rs.num_basis ? p0[kq] : 0.f;
  const float p1k = (p1 && kq < a.rs.num_basis) ? p1[kq] : 0.f;
  const float wk = (float)((double)p0k * 0.15915494309189535);  // p0 / (2 pi)
  if (t < 2) {
    int64_t target = (a.n_edges * (int64_t)(blockIdx.x + t) + gridDim.x - 1) / gridDim.x;
    int32_t n = (blockIdx.x + t == 0) ? 0 : ((blockIdx.x + t == gridDim.x) ? (int32_t)a.n_nodes : lower_node(a.rowptr, a.n_nodes, target));
    sm.range[t] = n;
  }
  __syncthreads();
  const int64_t n0 = sm.range[0], n1 = sm.range[1];
  if (n0 >= n1) return;
  const int32_t e_begin = a.rowptr[n0], e_end = a.rowptr[n1];

  int64_t cur = n0;
  float hs, he, hm, xh[5];
  float acc_hs = 0.f, acc_he = 0.f, acc_hm = 0.f, acc_xh[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  auto begin_node = [&](int64_t n) {
    hs = has_u ? h[n * H + t] : 0.f;
    he = has_u ? h[n * H + C + t] : 0.f;
    hm = has_s ? h[n * H + 2 * C + t] : 0.f;
#pragma unroll
    for (int m = 0; m < 5; ++m) xh[m] = (m < nm) ? xhat[xa.off + n * xa.node + m * xa.comp] : 0.f;
  };
  auto flush = [&](int64_t n) {
    if (has_u) {
      grad_h[n * H + t] = acc_hs;
      grad_h[n * H + C + t] = acc_he;
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < nm) grad_xhat[xa.off + n * xa.node + m * xa.comp] = acc_xh[m];
    }
    if (has_s) grad_h[n * H + 2 * C + t] = acc_hm;
    acc_hs = acc_he = acc_hm = 0.f;
#pragma unroll
    for (int m = 0; m < 5; ++m) acc_xh[m] = 0.f;
  };
  begin_node(cur);

  for (int32_t base = e_begin; base < e_end; base += TE) {
    const int cnt = min(TE, e_end - base);
    __syncthreads();
    phase0<KS, true>(a, vec, p0, p1, base, cnt, n0, n1, sm, wk, p0k, p1k);
    __syncthreads();
    if (!(a.ablate & 2)) phase1<KS, true>(a, wa, sh_bias, sm, sh_phi, sh_dphi);
    __syncthreads();
    if (a.ablate & 4) continue;
    // ---- phase 2
    float gx[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dgm = 0.f;
    auto gather = [&](int j, float (&ggx)[5], float& gdgm) {
      const int64_t c = sm.other[j];
      if (has_s) gdgm = grad_s[c * F + t];
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < nm) ggx[m] = grad_x[c * D + off + m];
    };
    gather(0, gx, dgm);
    for (int j = 0; j < cnt; ++j) {
      float ngx[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, ndgm = 0.f;
      if (j + 1 < cnt) gather(j + 1, ngx, ndgm);
      const int64_t dst = sm.self[j];
      while (cur < dst) {
        flush(cur);
        ++cur;
        begin_node(cur);
      }
      const float* ph = sh_phi + j * HP;
      const float* dph = sh_dphi + j * HP;
      const float ps = has_u ? ph[t] : 0.f, pe = has_u ? ph[C + t] : 0.f, pm = has_s ? ph[2 * C + t] : 0.f;
      const float qs = has_u ? dph[t] : 0.f, qe = has_u ? dph[C + t] : 0.f, qm = has_s ? dph[2 * C + t] : 0.f;
      float dgs = 0.f, dge = 0.f;
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const float yv = (m < nm) ? sm.y[j][yoff + m] : 0.f;
        dgs += xh[m] * gx[m];
        dge += yv * gx[m];
      }
      acc_hs += ps * dgs;
      acc_he += pe * dge;
      acc_hm += pm * dgm;
      const float gate = hs * ps;
#pragma unroll
      for (int m = 0; m < 5; ++m) acc_xh[m] += gate * gx[m];
      // per-edge scalars reduced over the channels of this wave
      const float pd = wave_sum_to_lane63(hs * dgs * qs + he * dge * qe + hm * dgm * qm);
      const float gy = he * pe;
      float r1[3] = {0.f, 0.f, 0.f}, r2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      if (wave_has1) {
#pragma unroll
        for (int m = 0; m < 3; ++m) r1[m] = wave_sum_to_lane63(l == 1 ? gy * gx[m] : 0.f);
      }
      if (wave_has2) {
#pragma unroll
        for (int m = 0; m < 5; ++m) r2[m] = wave_sum_to_lane63(l == 2 ? gy * gx[m] : 0.f);
      }
      if (lane == 63) {
        sm.red[j][wave][0] = pd;
#pragma unroll
        for (int m = 0; m < 3; ++m) sm.red[j][wave][1 + m] = r1[m];
#pragma unroll
        for (int m = 0; m < 5; ++m) sm.red[j][wave][4 + m] = r2[m];
      }
#pragma unroll
      for (int m = 0; m < 5; ++m) gx[m] = ngx[m];
      dgm = ndgm;
    }
    __syncthreads();
    if (t < cnt) {
      float gd = 0.f, q1[3] = {0.f, 0.f, 0.f}, q2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        gd += sm.red[t][w][0];
#pragma unroll
        for (int m = 0; m < 3; ++m) q1[m] += sm.red[t][w][1 + m];
#pragma unroll
        for (int m = 0; m < 5; ++m) q2[m] += sm.red[t][w][4 + m];
      }
      EdgeGeom<float> g;
      g.x = sm.g[t][0];
      g.y = sm.g[t][1];
      g.z = sm.g[t][2];
      g.d = sm.g[t][3];
      g.inv_d = sm.g[t][4];
      float out[3];
      edge_grad<float>(g, gd, q1, q2, out);
      const int64_t e = sm.eid[t];
      grad_vec[3 * e] = out[0];
      grad_vec[3 * e + 1] = out[1];
      grad_vec[3 * e + 2] = out[2];
    }
  }
  while (cur < n1) {
    flush(cur);
    ++cur;
    if (cur < n1) begin_node(cur);
  }
}

// host side ---------------------------------------------------------------------
bool mfma_path_supported(int dtype, int num_basis, int node_dim, const int32_t mul[3]) {
  const int C = mul[0] + mul[1] + mul[2];
  const int H = node_dim + 2 * C;
  return dtype == XEQ_F32 && num_basis <= 32 && C <= 256 && node_dim <= 256 && H <= 4 * TPW * 16;
}

void fill_msg2_args(Msg2Args& a, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                      const int64_t* other_idx, int rbf_kind, int cutoff_kind, int num_basis,
                      double cutoff, int node_dim, const int32_t mul[3]) {
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.rowptr = rowptr;
  a.perm = perm;
  a.other_idx = other_idx;
  for (int l = 0; l < 3; ++l) a.ir.mul[l] = mul[l];
  a.C = a.ir.C();
  a.D = a.ir.D();
  a.F = node_dim;
  a.H = a.F + 2 * a.C;
  a.NT = (a.H + 15) / 16;
  a.HP = a.NT * 16;
  while (a.HP % 32 != 4) a.HP += 4;  // row stride = 4 (mod 32) dwords: conflict-free ds_write_b128
  a.rs = RadialSpec{rbf_kind, cutoff_kind, num_basis, cutoff};
  const char* ab = getenv("XEQ_ABLATE");
  a.ablate = ab ? atoi(ab) : 0;
}

template <int KS>
static void launch_fwd(const Msg2Args& a, unsigned grid, hipStream_t st, const float* vec, const float* h, const float* xhat,
                       const float* s_in, const float* x_in, const float* w, const float* b, const float* p0,
                       const float* p1, float* s_out, float* x_out) {
  size_t dyn = sizeof(float) * (size_t)(TE * a.HP + a.NT * 16);
  hipLaunchKernelGGL((k_message_fwd_mfma<KS>), dim3(grid), dim3(256), dyn, st, a, vec, h, xhat, s_in, x_in, w, b, p0, p1,
                     s_out, x_out);
}

template <int KS>
static void launch_bwd(const Msg2Args& a, unsigned grid, hipStream_t st, const float* vec, const float* h, const float* xhat,
                       const float* gs, const float* gx, const float* w, const float* b, const float* p0, const float* p1,
                       float* gh, float* gxh, float* gv) {
  size_t dyn = sizeof(float) * (size_t)(2 * TE * a.HP + a.NT * 16);
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_message_bwd_mfma<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_message_bwd_mfma<KS>), dim3(grid), dim3(256), dyn, st, a, vec, h, xhat, gs, gx, w, b, p0, p1, gh,
                     gxh, gv);
}

int message_fwd_mfma(int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                     const int64_t* nbr, const void* vec, const void* h, const void* xhat, const void* s_in,
                     const void* x_in, const void* w_rbf, const void* b_rbf, const void* p0, const void* p1, int rbf_kind,
                     int cutoff_kind, int num_basis, double cutoff, int node_dim, const int32_t mul[3], void* s_out,
                     void* x_out, int xl, void* stream) {
  Msg2Args a{};
  fill_msg2_args(a, n_nodes, n_edges, rowptr, perm, nbr, rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul);
  a.xl = xl;
  int64_t want = 256 * 2;  // resident workgroups: 2 per CU at ~170 VGPRs (one full wave of blocks, no tail round)
  unsigned grid = (unsigned)(n_nodes < want ? n_nodes : want);
  hipStream_t st = (hipStream_t)stream;
#define XEQ_ARGS_F (const float*)vec, (const float*)h, (const float*)xhat, (const float*)s_in, (const float*)x_in, \
                   (const float*)w_rbf, (const float*)b_rbf, (const float*)p0, (const float*)p1, (float*)s_out, (float*)x_out
  if (num_basis <= 8) launch_fwd<2>(a, grid, st, XEQ_ARGS_F);
  else if (num_basis <= 16) launch_fwd<4>(a, grid, st, XEQ_ARGS_F);
  else if (num_basis <= 20) launch_fwd<5>(a, grid, st, XEQ_ARGS_F);
  else launch_fwd<8>(a, grid, st, XEQ_ARGS_F);
#undef XEQ_ARGS_F
  return XEQ_OK;
}

int message_bwd_mfma(int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                     const int64_t* center, const void* vec, const void* h, const void* xhat,
                     const void* grad_s, const void* grad_x, const void* w_rbf, const void* b_rbf, const void* p0,
                     const void* p1, int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                     const int32_t mul[3], void* grad_h, void* grad_xhat, void* grad_vec, int xl, void* stream) {
  Msg2Args a{};
  fill_msg2_args(a, n_nodes, n_edges, n_rowptr, n_perm, center, rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul);
  a.xl = xl;
  int64_t want = 256 * 2;
  unsigned grid = (unsigned)(n_nodes < want ? n_nodes : want);
  hipStream_t st = (hipStream_t)stream;
#define XEQ_ARGS_B (const float*)vec, (const float*)h, (const float*)xhat, (const float*)grad_s, (const float*)grad_x, \
                   (const float*)w_rbf, (const float*)b_rbf, (const float*)p0, (const float*)p1, (float*)grad_h,      \
                   (float*)grad_xhat, (float*)grad_vec
  if (num_basis <= 8) launch_bwd<2>(a, grid, st, XEQ_ARGS_B);
  else if (num_basis <= 16) launch_bwd<4>(a, grid, st, XEQ_ARGS_B);
  else if (num_basis <= 20) launch_bwd<5>(a, grid, st, XEQ_ARGS_B);
  else launch_bwd<8>(a, grid, st, XEQ_ARGS_B);
#undef XEQ_ARGS_B
  return XEQ_OK;
}

}  // namespace xeq
