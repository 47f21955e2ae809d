// Fused XPaiNN message kernels, LDS-window form for graphs made of small CLOSED node
// segments (molecule batches: no edge leaves its molecule).  Reference dataflow:
// nn/xpainn.py:140-159; reverse pass for nn/basic.py:143-159.
//
// Both directions walk the CSR over NEIGHBORS (source-stationary):
//   forward   the source node's rows h[n], xhat[n] sit in registers (read ONCE from HBM,
//             coalesced, prefetched one node ahead); every out-edge (n -> c) adds its message
//             into the destination accumulators of the segment, which live in LDS
//             (acc[c - a][F + D]); thread t owns channel t of every row, so there are no
//             races and no atomics, and the summation order is fixed => bitwise reproducible;
//             the segment's rows are written once: s_out = s_in + acc, x_out = x_in + acc.
//   reverse   the segment's grad_s / grad_x rows are staged in LDS once; the source node's
//             rows stay in registers; grad_h / grad_xhat are register sums written once per
//             node; per-edge dL/dd, dL/dY_lm are reduced with DPP and written as grad_vec.
// HBM traffic is therefore the algorithmic minimum (every node row is read once per launch)
// instead of one 4.2 kB row gather per edge.  Tile phases 0/1 (radial terms, rbf_lin on the
// matrix cores) are those of xeq_message_mfma.hip; phase-0 inputs of the NEXT tile are
// prefetched into registers while the current tile is processed.
#include "xeq_message_tile.h"

namespace xeq {

struct SegArgs {
  Msg2Args m;               // rowptr = CSR over neighbors, other_idx unused
  const int32_t* seg_ptr;   // [S+1] node boundaries of the closed segments
  const int32_t* seg_eptr;  // [S+1] = rowptr[seg_ptr]
  int n_seg;
  int WN;                   // window rows available in LDS (>= largest segment)
  const float* vec_n;       // [E,3] edge vectors in slot order
  const int32_t* other_n;   // [E]  destination (center) node of each slot
  const int32_t* eid_n;     // [E]  edge id of each slot (n_perm)
};

// registers that carry the next tile's phase-0 inputs
struct Pre {
  float v[2][3];   // rho lanes: vectors of edges eh, eh + 8
  float rv[3];     // record lanes
  int32_t other, eid;
};

template <int KS>
__device__ __forceinline__ void prefetch_tile(const SegArgs& a, int32_t base, int cnt, Pre& pf) {
  constexpr int KB = 4 * KS;
  const int t = threadIdx.x;
  if (t < 8 * KB) {
    const int eh = t / KB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = eh + 8 * half;
      if (j < cnt) {
        const float* v = a.vec_n + 3 * (int64_t)(base + j);
        pf.v[half][0] = v[0];
        pf.v[half][1] = v[1];
        pf.v[half][2] = v[2];
      }
    }
  }
  const int j = t - (KB == 32 ? 240 : 192);
  if (j >= 0 && j < cnt) {
    const float* v = a.vec_n + 3 * (int64_t)(base + j);
    pf.rv[0] = v[0];
    pf.rv[1] = v[1];
    pf.rv[2] = v[2];
    pf.other = a.other_n[base + j];
    pf.eid = a.eid_n[base + j];
  }
}

// node that owns slot p, searched in the LDS copy of rowptr[a .. a+nw]
__device__ __forceinline__ int32_t node_of_slot_lds(const int32_t* __restrict__ sh_rowptr, int nw, int32_t p) {
  int lo = 0, hi = nw;  // first i in [0, nw) with sh_rowptr[i + 1] > p
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (sh_rowptr[mid + 1] > p) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

template <int KS, bool BWD>
__device__ __forceinline__ void phase0_regs(const SegArgs& a, const Pre& pf, int32_t base, int cnt,
                                            const int32_t* __restrict__ sh_rowptr, int nw, int32_t node_a,
                                            Smem<KS, BWD>& sm, float wk, float p0k, float p1k) {
  constexpr int KB = 4 * KS;
  const float rc = (float)a.m.rs.cutoff;
  const int t = threadIdx.x;
  if (t < 8 * KB) {
    const int k = t % KB, eh = t / KB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int j = eh + 8 * half;
      float r = 0.f, dr = 0.f;
      if (j < cnt && k < a.m.rs.num_basis) {
        const float vx = pf.v[half][0], vy = pf.v[half][1], vz = pf.v[half][2];
        const float d = sqrtf(vx * vx + vy * vy + vz * vz);
        radial_fast(a.m.rs.rbf_kind, d, rc, p0k, p1k, wk, r, dr);
      }
      sm.rho[j][k] = r;
      if (BWD) sm.drho[j][k] = dr;
    }
  }
  const int j = t - (KB == 32 ? 240 : 192);
  if (j >= 0 && j < TE) {
    if (j < cnt) {
      EdgeGeom<float> g = edge_geom<float>(pf.rv[0], pf.rv[1], pf.rv[2]);
      float f, df, y1[3], y2[5];
      envelope_fast(a.m.rs.cutoff_kind, g.d, rc, f, df);
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
      sm.self[j] = node_a + node_of_slot_lds(sh_rowptr, nw, base + j);
      sm.other[j] = pf.other;
      sm.eid[j] = pf.eid;
    } else {
      sm.y[j][9] = 0.f;
      sm.y[j][10] = 0.f;
    }
  }
}

// equal-edge split of the segment list over the grid: first segment with seg_eptr >= target
__device__ __forceinline__ int first_segment(const SegArgs& a, int b) {
  if (b <= 0) return 0;
  if (b >= (int)gridDim.x) return a.n_seg;
  const int64_t target = (a.m.n_edges * (int64_t)b + gridDim.x - 1) / gridDim.x;
  int lo = 0, hi = a.n_seg;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a.seg_eptr[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

template <int KS>
__global__ void __launch_bounds__(256) k_message_fwd_seg(SegArgs a, const float* __restrict__ h,
                                                         const float* __restrict__ xhat, const float* __restrict__ s_in,
                                                         const float* __restrict__ x_in, const float* __restrict__ w_rbf,
                                                         const float* __restrict__ b_rbf, const float* __restrict__ p0,
                                                         const float* __restrict__ p1, float* __restrict__ s_out,
                                                         float* __restrict__ x_out) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ Smem<KS, false> sm;
  const Msg2Args& ma = a.m;
  const int C = ma.C, F = ma.F, D = ma.D, H = ma.H, HP = ma.HP, AW = F + D;
  float* sh_phi = dyn;                          // [TE][HP]
  float* sh_acc = dyn + TE * HP;                // [WN][AW]
  int32_t* sh_rowptr = reinterpret_cast<int32_t*>(sh_acc + a.WN * AW);  // [WN + 1]
  float* sh_bias = reinterpret_cast<float*>(sh_rowptr + ((a.WN + 1 + 3) & ~3));  // [NT*16], 16-B aligned
  const int t = threadIdx.x;
  const bool has_u = t < C, has_s = t < F;
  int l = 0, off = 0;
  if (has_u) ma.ir.locate(t, l, off);
  const int nm = has_u ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const XAddr xa = xaddr(ma.ir, ma.n_nodes, has_u ? t : 0, ma.xl);

  float wa[TPW][KS];
  load_a_frags<KS>(ma, w_rbf, wa);
  for (int c = t; c < ma.NT * 16; c += 256) sh_bias[c] = c < ma.H ? b_rbf[c] : 0.f;
  const int kq = t % (4 * KS);
  const float p0k = kq < ma.rs.num_basis ? p0[kq] : 0.f;
  const float p1k = (p1 && kq < ma.rs.num_basis) ? p1[kq] : 0.f;
  const float wk = (float)((double)p0k * 0.15915494309189535);
  if (t < 2) sm.range[t] = first_segment(a, blockIdx.x + t);
  __syncthreads();
  const int seg0 = sm.range[0], seg1 = sm.range[1];

  for (int seg = seg0; seg < seg1; ++seg) {
    const int32_t na = a.seg_ptr[seg], nb = a.seg_ptr[seg + 1];
    const int nw = nb - na;
    const int32_t e_begin = a.seg_eptr[seg], e_end = a.seg_eptr[seg + 1];
    __syncthreads();  // previous segment written out
    for (int i = t; i < nw * AW; i += 256) sh_acc[i] = 0.f;
    for (int i = t; i <= nw; i += 256) sh_rowptr[i] = ma.rowptr[na + i];
    Pre pf;
    prefetch_tile<KS>(a, e_begin, min(TE, e_end - e_begin), pf);
    // source rows: current node in registers, next node prefetched
    int32_t cur = na;
    float hs = 0.f, he = 0.f, hm = 0.f, xh[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float nhs = 0.f, nhe = 0.f, nhm = 0.f, nxh[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_row = [&](int64_t n, float& rs, float& re, float& rm, float (&rx)[5]) {
      if (has_u) {
        rs = h[n * H + t];
        re = h[n * H + C + t];
#pragma unroll
        for (int m = 0; m < 5; ++m)
          if (m < nm) rx[m] = xhat[xa.off + n * xa.node + m * xa.comp];
      }
      if (has_s) rm = h[n * H + 2 * C + t];
    };
    if (nw > 0) load_row(na, hs, he, hm, xh);
    if (nw > 1) load_row(na + 1, nhs, nhe, nhm, nxh);

    for (int32_t base = e_begin; base < e_end; base += TE) {
      const int cnt = min(TE, e_end - base);
      __syncthreads();  // previous tile consumed (and the zero-fill / rowptr copy on the first tile)
      phase0_regs<KS, false>(a, pf, base, cnt, sh_rowptr, nw, na, sm, wk, p0k, p1k);
      if (base + TE < e_end) prefetch_tile<KS>(a, base + TE, min(TE, e_end - base - TE), pf);
      __syncthreads();
      phase1<KS, false>(ma, wa, sh_bias, sm, sh_phi, nullptr);
      __syncthreads();
      // ---- phase 2: scatter this tile's messages into the LDS accumulators
      for (int j = 0; j < cnt; ++j) {
        const int32_t src = sm.self[j];
        while (cur < src) {  // uniform: rotate the prefetched row in, start the next prefetch
          ++cur;
          hs = nhs;
          he = nhe;
          hm = nhm;
#pragma unroll
          for (int m = 0; m < 5; ++m) xh[m] = nxh[m];
          if (cur + 1 < nb) load_row(cur + 1, nhs, nhe, nhm, nxh);
        }
        const int ci = sm.other[j] - na;
        float* acc = sh_acc + ci * AW;
        const float* ph = sh_phi + j * HP;
        if (has_s) acc[t] += hm * ph[2 * C + t];
        if (has_u) {
          const float gs = hs * ph[t], ge = he * ph[C + t];
#pragma unroll
          for (int m = 0; m < 5; ++m)
            if (m < nm) acc[F + off + m] += xh[m] * gs + sm.y[j][yoff + m] * ge;
        }
      }
    }
    __syncthreads();
    // ---- write the segment: out = in + acc (coalesced rows)
    for (int i = t; i < nw * F; i += 256) {
      const int r = i / F, c = i - r * F;
      const int64_t g = (int64_t)(na + r) * F + c;
      s_out[g] = s_in[g] + sh_acc[r * AW + c];
    }
    for (int i = t; i < nw * D; i += 256) {
      const int r = i / D, c = i - r * D;
      const int64_t g = (int64_t)(na + r) * D + c;
      x_out[g] = x_in[g] + sh_acc[r * AW + F + c];
    }
  }
}

template <int KS>
__global__ void __launch_bounds__(256) k_message_bwd_seg(SegArgs a, const float* __restrict__ h,
                                                         const float* __restrict__ xhat, const float* __restrict__ grad_s,
                                                         const float* __restrict__ grad_x, const float* __restrict__ w_rbf,
                                                         const float* __restrict__ b_rbf, const float* __restrict__ p0,
                                                         const float* __restrict__ p1, float* __restrict__ grad_h,
                                                         float* __restrict__ grad_xhat, float* __restrict__ grad_vec) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ Smem<KS, true> sm;
  const Msg2Args& ma = a.m;
  const int C = ma.C, F = ma.F, D = ma.D, H = ma.H, HP = ma.HP, AW = F + D;
  float* sh_phi = dyn;                          // [TE][HP]
  float* sh_dphi = dyn + TE * HP;               // [TE][HP]
  float* sh_g = dyn + 2 * TE * HP;              // [WN][AW]  grad_s | grad_x rows of the segment
  int32_t* sh_rowptr = reinterpret_cast<int32_t*>(sh_g + a.WN * AW);
  float* sh_bias = reinterpret_cast<float*>(sh_rowptr + ((a.WN + 1 + 3) & ~3));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool has_u = t < C, has_s = t < F;
  int l = 0, off = 0;
  if (has_u) ma.ir.locate(t, l, off);
  const int nm = has_u ? 2 * l + 1 : 0;
  const int yoff = l == 0 ? 0 : (l == 1 ? 1 : 4);
  const bool wave_has1 = __ballot(has_u && l == 1) != 0ull;
  const bool wave_has2 = __ballot(has_u && l == 2) != 0ull;
  const XAddr xa = xaddr(ma.ir, ma.n_nodes, has_u ? t : 0, ma.xl);

  float wa[TPW][KS];
  load_a_frags<KS>(ma, w_rbf, wa);
  for (int c = t; c < ma.NT * 16; c += 256) sh_bias[c] = c < ma.H ? b_rbf[c] : 0.f;
  const int kq = t % (4 * KS);
  const float p0k = kq < ma.rs.num_basis ? p0[kq] : 0.f;
  const float p1k = (p1 && kq < ma.rs.num_basis) ? p1[kq] : 0.f;
  const float wk = (float)((double)p0k * 0.15915494309189535);
  if (t < 2) sm.range[t] = first_segment(a, blockIdx.x + t);
  __syncthreads();
  const int seg0 = sm.range[0], seg1 = sm.range[1];

  for (int seg = seg0; seg < seg1; ++seg) {
    const int32_t na = a.seg_ptr[seg], nb = a.seg_ptr[seg + 1];
    const int nw = nb - na;
    const int32_t e_begin = a.seg_eptr[seg], e_end = a.seg_eptr[seg + 1];
    __syncthreads();  // previous segment's tiles consumed
    for (int i = t; i < nw * F; i += 256) {
      const int r = i / F, c = i - r * F;
      sh_g[r * AW + c] = grad_s[(int64_t)(na + r) * F + c];
    }
    for (int i = t; i < nw * D; i += 256) {
      const int r = i / D, c = i - r * D;
      sh_g[r * AW + F + c] = grad_x[(int64_t)(na + r) * D + c];
    }
    for (int i = t; i <= nw; i += 256) sh_rowptr[i] = ma.rowptr[na + i];
    Pre pf;
    prefetch_tile<KS>(a, e_begin, min(TE, e_end - e_begin), pf);
    int32_t cur = na;
    float hs = 0.f, he = 0.f, hm = 0.f, xh[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float nhs = 0.f, nhe = 0.f, nhm = 0.f, nxh[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_row = [&](int64_t n, float& rs, float& re, float& rm, float (&rx)[5]) {
      if (has_u) {
        rs = h[n * H + t];
        re = h[n * H + C + t];
#pragma unroll
        for (int m = 0; m < 5; ++m)
          if (m < nm) rx[m] = xhat[xa.off + n * xa.node + m * xa.comp];
      }
      if (has_s) rm = h[n * H + 2 * C + t];
    };
    if (nw > 0) load_row(na, hs, he, hm, xh);
    if (nw > 1) load_row(na + 1, nhs, nhe, nhm, nxh);
    float acc_hs = 0.f, acc_he = 0.f, acc_hm = 0.f, acc_xh[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
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

    for (int32_t base = e_begin; base < e_end; base += TE) {
      const int cnt = min(TE, e_end - base);
      __syncthreads();
      phase0_regs<KS, true>(a, pf, base, cnt, sh_rowptr, nw, na, sm, wk, p0k, p1k);
      if (base + TE < e_end) prefetch_tile<KS>(a, base + TE, min(TE, e_end - base - TE), pf);
      __syncthreads();
      phase1<KS, true>(ma, wa, sh_bias, sm, sh_phi, sh_dphi);
      __syncthreads();
      for (int j = 0; j < cnt; ++j) {
        const int32_t src = sm.self[j];
        while (cur < src) {
          flush(cur);
          ++cur;
          hs = nhs;
          he = nhe;
          hm = nhm;
#pragma unroll
          for (int m = 0; m < 5; ++m) xh[m] = nxh[m];
          if (cur + 1 < nb) load_row(cur + 1, nhs, nhe, nhm, nxh);
        }
        const float* grow = sh_g + (sm.other[j] - na) * AW;
        const float* ph = sh_phi + j * HP;
        const float* dph = sh_dphi + j * HP;
        const float ps = has_u ? ph[t] : 0.f, pe = has_u ? ph[C + t] : 0.f, pm = has_s ? ph[2 * C + t] : 0.f;
        const float qs = has_u ? dph[t] : 0.f, qe = has_u ? dph[C + t] : 0.f, qm = has_s ? dph[2 * C + t] : 0.f;
        const float dgm = has_s ? grow[t] : 0.f;
        float gx[5], dgs = 0.f, dge = 0.f;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          gx[m] = (m < nm) ? grow[F + off + m] : 0.f;
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
    // remaining nodes of the segment (incl. nodes without out-edges)
    while (cur < nb) {
      flush(cur);
      ++cur;
      if (cur < nb) {
        hs = nhs;
        he = nhe;
        hm = nhm;
#pragma unroll
        for (int m = 0; m < 5; ++m) xh[m] = nxh[m];
        if (cur + 1 < nb) load_row(cur + 1, nhs, nhe, nhm, nxh);
      }
    }
  }
}

// ------------------------------------------------------------------------- host side
static constexpr size_t LDS_BYTES = 160 * 1024;

static size_t seg_lds_bytes(const Msg2Args& m, int WN, bool bwd) {
  const size_t AW = (size_t)m.F + m.D;
  // [phi (, dphi)] [window rows] [rowptr copy, 16-B padded] [bias]
  const size_t wn_pad = ((size_t)WN + 1 + 3) / 4 * 4;
  return sizeof(float) * ((bwd ? 2 : 1) * (size_t)TE * m.HP + (size_t)WN * AW + (size_t)m.NT * 16) + sizeof(int32_t) * wn_pad;
}

void fill_msg2_args(Msg2Args& a, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                    const int64_t* other_idx, int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                    const int32_t mul[3]);

// largest segment the LDS window can hold for this configuration (0 = path unusable)
int seg_path_max_nodes(int dtype, int num_basis, int node_dim, const int32_t mul[3], int bwd) {
  if (!mfma_path_supported(dtype, num_basis, node_dim, mul)) return 0;
  Msg2Args m{};
  fill_msg2_args(m, 0, 0, nullptr, nullptr, nullptr, 0, 0, num_basis, 1.0, node_dim, mul);
  const size_t stat = bwd ? 8192 : 4096;  // static Smem + slack
  const size_t tiles = sizeof(float) * (bwd ? 2 : 1) * (size_t)TE * m.HP;
  if (LDS_BYTES < stat + tiles + 64) return 0;
  const size_t bias = sizeof(float) * (size_t)m.NT * 16;
  if (LDS_BYTES < stat + tiles + bias + 256) return 0;
  return (int)((LDS_BYTES - stat - tiles - bias - 256) / (sizeof(float) * ((size_t)m.F + m.D) + 4));
}

template <int KS>
static int launch_fwd_seg(const SegArgs& a, unsigned grid, size_t dyn, hipStream_t st, const float* h, const float* xhat,
                          const float* s_in, const float* x_in, const float* w, const float* b, const float* p0,
                          const float* p1, float* s_out, float* x_out) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_message_fwd_seg<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BYTES - 4096));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_message_fwd_seg<KS>), dim3(grid), dim3(256), dyn, st, a, h, xhat, s_in, x_in, w, b, p0, p1, s_out, x_out);
  return XEQ_OK;
}

template <int KS>
static int launch_bwd_seg(const SegArgs& a, unsigned grid, size_t dyn, hipStream_t st, const float* h, const float* xhat,
                          const float* gs, const float* gx, const float* w, const float* b, const float* p0, const float* p1,
                          float* gh, float* gxh, float* gv) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_message_bwd_seg<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BYTES - 8192));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_message_bwd_seg<KS>), dim3(grid), dim3(256), dyn, st, a, h, xhat, gs, gx, w, b, p0, p1, gh, gxh, gv);
  return XEQ_OK;
}

static void fill_seg(SegArgs& a, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* seg_ptr,
                     const int32_t* seg_eptr, int n_seg, int max_seg, const void* vec_n, const int32_t* other_n,
                     const int32_t* eid_n, int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                     const int32_t mul[3]) {
  fill_msg2_args(a.m, n_nodes, n_edges, n_rowptr, nullptr, nullptr, rbf_kind, cutoff_kind, num_basis, cutoff, node_dim, mul);
  a.seg_ptr = seg_ptr;
  a.seg_eptr = seg_eptr;
  a.n_seg = n_seg;
  a.WN = max_seg;
  a.vec_n = (const float*)vec_n;
  a.other_n = other_n;
  a.eid_n = eid_n;
}

static unsigned seg_grid(int n_seg, size_t dyn, size_t stat) {
  int per_cu = (int)(LDS_BYTES / (dyn + stat));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int64_t g = 256ll * per_cu;
  return (unsigned)(n_seg < g ? n_seg : g);
}

int message_fwd_seg(int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* seg_ptr,
                    const int32_t* seg_eptr, int n_seg, int max_seg, const void* vec_n, const int32_t* other_n,
                    const int32_t* eid_n, const void* h, const void* xhat, const void* s_in, const void* x_in,
                    const void* w_rbf, const void* b_rbf, const void* p0, const void* p1, int rbf_kind, int cutoff_kind,
                    int num_basis, double cutoff, int node_dim, const int32_t mul[3], void* s_out, void* x_out,
                    int xl, void* stream) {
  SegArgs a{};
  fill_seg(a, n_nodes, n_edges, n_rowptr, seg_ptr, seg_eptr, n_seg, max_seg, vec_n, other_n, eid_n, rbf_kind, cutoff_kind,
           num_basis, cutoff, node_dim, mul);
  a.m.xl = xl;
  const size_t dyn = seg_lds_bytes(a.m, a.WN, false);
  const unsigned grid = seg_grid(n_seg, dyn, 4096);
  hipStream_t st = (hipStream_t)stream;
#define XEQ_A (const float*)h, (const float*)xhat, (const float*)s_in, (const float*)x_in, (const float*)w_rbf, \
              (const float*)b_rbf, (const float*)p0, (const float*)p1, (float*)s_out, (float*)x_out
  if (num_basis <= 8) return launch_fwd_seg<2>(a, grid, dyn, st, XEQ_A);
  if (num_basis <= 16) return launch_fwd_seg<4>(a, grid, dyn, st, XEQ_A);
  if (num_basis <= 20) return launch_fwd_seg<5>(a, grid, dyn, st, XEQ_A);
  return launch_fwd_seg<8>(a, grid, dyn, st, XEQ_A);
#undef XEQ_A
}

int message_bwd_seg(int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* seg_ptr,
                    const int32_t* seg_eptr, int n_seg, int max_seg, const void* vec_n, const int32_t* other_n,
                    const int32_t* eid_n, const void* h, const void* xhat, const void* grad_s, const void* grad_x,
                    const void* w_rbf, const void* b_rbf, const void* p0, const void* p1, int rbf_kind, int cutoff_kind,
                    int num_basis, double cutoff, int node_dim, const int32_t mul[3], void* grad_h, void* grad_xhat,
                    void* grad_vec, int xl, void* stream) {
  SegArgs a{};
  fill_seg(a, n_nodes, n_edges, n_rowptr, seg_ptr, seg_eptr, n_seg, max_seg, vec_n, other_n, eid_n, rbf_kind, cutoff_kind,
           num_basis, cutoff, node_dim, mul);
  a.m.xl = xl;
  const size_t dyn = seg_lds_bytes(a.m, a.WN, true);
  const unsigned grid = seg_grid(n_seg, dyn, 8192);
  hipStream_t st = (hipStream_t)stream;
#define XEQ_A (const float*)h, (const float*)xhat, (const float*)grad_s, (const float*)grad_x, (const float*)w_rbf, \
              (const float*)b_rbf, (const float*)p0, (const float*)p1, (float*)grad_h, (float*)grad_xhat, (float*)grad_vec
  if (num_basis <= 8) return launch_bwd_seg<2>(a, grid, dyn, st, XEQ_A);
  if (num_basis <= 16) return launch_bwd_seg<4>(a, grid, dyn, st, XEQ_A);
  if (num_basis <= 20) return launch_bwd_seg<5>(a, grid, dyn, st, XEQ_A);
  return launch_bwd_seg<8>(a, grid, dyn, st, XEQ_A);
#undef XEQ_A
}

}  // namespace xeq
