// Fused XPaiNN message kernels, "wave / matrix-core" form (fp32; default).
// Reference dataflow: nn/xpainn.py:140-159; reverse pass for nn/basic.py:143-159.
//
// What bounds the op: per edge and layer the filter  phi_e = (W rho(d_e) + b) f(d_e)  is a
// [576 x 21] x [21] contraction -- 24 kFLOP against 4.2 kB of gathered node rows -- and every other
// quantity is a handful of FMAs per channel.  This form puts the contraction on the matrix cores as
// exact-f32 MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain) and keeps the aggregation in registers:
//
//   * a wave owns (range of nodes, unit); a unit is 32 gate channels of one l together with their
//     scalar-message channels (l = 0): 4 + 2 + 1 = 7 units for 128x0e+64x1o+32x2e.  Waves are independent:
//     no barriers, no shared accumulators, no atomics.
//   * a tile is D[edge][channel] = rho~[edge][k] x W~[k][channel], K = B + 1 (bias column times the envelope):
//     the CHANNEL is on the lane, the 16 accumulator registers of a lane are 16 consecutive edges.  The weights
//     are the B operand and stay in VGPRs for the whole launch; the A operand is the per-edge record written
//     once per evaluation by k_edge_basis_wm (shared by the 3 layers and both directions).
//   * the two half-waves are two independent STREAMS: each walks its own contiguous range of CSR segments
//     (forward: edges sorted by center; reverse: sorted by neighbor), 16 edges per tile, so the sum over the
//     edges of a node is a running sum in the lane's registers across rows and tiles, stored once when the
//     segment ends: deterministic, written exactly once, no read-modify-write anywhere.
//   * per row the lanes of a half read 128 contiguous bytes of the gathered node row (coalesced, L1/L2
//     resident: a molecule's rows are re-read by every center of the molecule).
//   * the reverse pass needs, per edge, sums over the channels (dL/dd, dL/dY_lm): DPP row reductions per row,
//     written as per-unit partials [unit][E] and combined in fixed order by k_wm_edge_grad.
//
// Measured dead end, kept out of the code: accumulating with ds_add_f32 into LDS rows (edge on the lane) --
// LDS float atomics retire about one lane per 2.3 cycles on gfx950 (149 cycles per wave instruction measured),
// 870 us of a 1.19 ms forward launch.
#include "xeq_common.h"

namespace xeq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// record layout (floats): KS = ceil((B + 1) / 2) MFMA k-steps, KP = roundup(KS, 4)
//   [0, KP)        even k:  val(0), val(2), ...     val(k) = f rho_k (k < B) | f (k == B) | 0
//   [KP, 2 KP)     odd k:   val(1), val(3), ...
//   [2 KP, 2 KP+8) Y1[3], Y2[5]          (value record; unused in the derivative record)
__host__ __device__ inline int wm_ks(int B) { return (B + 2) / 2; }

__global__ void k_edge_basis_wm(const float* __restrict__ vec, int64_t E, RadialSpec rs, int KP, const float* __restrict__ p0,
                                const float* __restrict__ p1, float* __restrict__ rec, float* __restrict__ drec) {
  const int B = rs.num_basis, EW = 2 * KP + 8;  // KP of the dispatched KS bucket: val(k) = 0 for k > B
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * EW) return;
  const int64_t e = t / EW;
  const int slot = (int)(t - e * EW);
  const float rc = (float)rs.cutoff;
  const EdgeGeom<float> g = edge_geom<float>(vec[3 * e], vec[3 * e + 1], vec[3 * e + 2]);
  if (slot < 2 * KP) {
    const int k = slot < KP ? 2 * slot : 2 * (slot - KP) + 1;
    float f, df, v = 0.f, dv = 0.f;
    envelope<float>(rs.cutoff_kind, g.d, rc, f, df);
    if (k < B) {
      float rho, drho;
      radial<float>(rs.rbf_kind, g.d, rc, p0[k], p1 ? p1[k] : 0.f, rho, drho);
      v = f * rho;
      dv = df * rho + f * drho;
    } else if (k == B) {
      v = f;
      dv = df;
    }
    rec[t] = v;
    if (drec) drec[t] = dv;
  } else {
    float y1[3], y2[5];
    sph_harm_l12<float>(g, y1, y2);
    const int q = slot - 2 * KP;
    const float y[8] = {y1[0], y1[1], y1[2], y2[0], y2[1], y2[2], y2[3], y2[4]};
    float v = y[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = q == i ? y[i] : v;
    rec[t] = v;
    if (drec) drec[t] = 0.f;
  }
}

struct WmArgs {
  int64_t n_nodes, n_edges;
  int n_ranges;                // node ranges; range w = streams 2w (half-wave 0) and 2w + 1 (half-wave 1)
  const int32_t* stream_ptr;   // [2 n_ranges + 1] node boundaries of the streams
  const int32_t* rowptr;       // [N + 1] slots of the walk order per owner node
  const int32_t* slot_eid;     // [E] edge id per slot (NULL: identity)
  const int32_t* slot_owner;   // [E] node whose segment the slot belongs to (forward: center, reverse: neighbor)
  const int32_t* slot_gather;  // [E] node whose rows are gathered    (forward: neighbor, reverse: center)
  int F, C, D, H, B;
  Irreps ir;
  int xl;                      // layout of xhat / grad_xhat
  int nu[3];                   // 32-channel units per l
};

struct WmUnit {
  int l, cb;       // l and index of the 32-channel block inside l
  int u0;          // first gate channel
  int xbase;       // flat e3nn offset of channel u0 (x_in / x_out / grad_x rows)
};
__device__ __forceinline__ WmUnit wm_unit(const WmArgs& a, int u) {
  WmUnit w;
  if (u < a.nu[0]) {
    w.l = 0;
    w.cb = u;
  } else if (u < a.nu[0] + a.nu[1]) {
    w.l = 1;
    w.cb = u - a.nu[0];
  } else {
    w.l = 2;
    w.cb = u - a.nu[0] - a.nu[1];
  }
  const int cbase = w.l == 0 ? 0 : (w.l == 1 ? a.ir.mul[0] : a.ir.mul[0] + a.ir.mul[1]);
  w.u0 = cbase + 32 * w.cb;
  int l_, off;
  a.ir.locate(w.u0, l_, off);
  w.xbase = off;
  return w;
}

constexpr int WM_WAVES = 4;   // waves per workgroup (independent of one another)

// (range, unit) of this wave.  Consecutive work items are the units of one range (they share its edge records and
// index arrays); consecutive groups of WM_WAVES items are dealt to workgroups so that neighbours in that order run on
// the same XCD (blocks b and b + 8 share one under round-robin dispatch: speed only).
__device__ __forceinline__ bool wm_decode(const WmArgs& a, int nunits, int& range, int& unit) {
  const int nb = gridDim.x, b = blockIdx.x;
  const int vb = (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
  const int item = vb * WM_WAVES + (threadIdx.x >> 6);
  range = item / nunits;
  unit = item - range * nunits;
  return range < a.n_ranges;
}

// rbf_lin rows of the unit as the B operand: lane (j = lane & 31 -> channel row0 + j, kh = lane >> 5) holds
// W~[row][2 s + kh], s < KS, with W~[., B] = bias and zeros beyond.
template <int KS>
__device__ __forceinline__ void wm_load_weights(const float* __restrict__ w, const float* __restrict__ b, int row, int B,
                                                int kh, float (&W)[KS]) {
  const float bias = b[row];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + kh;
    const float wv = w[(int64_t)row * B + (k < B ? k : B - 1)];
    W[s] = k < B ? wv : (k == B ? bias : 0.f);
  }
}

// A operand of a tile: lane (i = lane & 31 -> row i, kh) holds rec[edge of row i][2 s + kh]
template <int KS>
__device__ __forceinline__ void wm_load_record(const float* __restrict__ rp, int kh, bool valid, float (&R)[KS]) {
  constexpr int KP = (KS + 3) & ~3;
  const f32x4* __restrict__ q = reinterpret_cast<const f32x4*>(rp + kh * KP);
#pragma unroll
  for (int i = 0; i < KP / 4; ++i) {
    const f32x4 v = q[i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * i + r < KS) R[4 * i + r] = valid ? v[r] : 0.f;
  }
}

template <int KS>
__device__ __forceinline__ f32x16 wm_filter(const float (&R)[KS], const float (&W)[KS]) {
  f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(R[s], W[s], d, 0, 0, 0);
  return d;
}

// The tile table (LDS, private to the wave): per position p = 16 * half + v (v-th edge of the half-wave's
// stream in this tile) the quantities every lane of the half needs for that row, written by the lane that owns
// MFMA row i(p) = 8 (v >> 2) + 4 half + (v & 3), read back with half-uniform addresses (broadcast reads).
enum { T_GOFF = 0, T_OWN = 1, T_FLAGS = 2, T_EID = 3, T_Y = 4, T_ROWS = 12 };
enum { WM_FIRST = 1, WM_LAST = 2, WM_VALID = 4 };

// Tile setup runs two tiles ahead of the arithmetic, in three steps whose loads are issued one tile apart:
//   wm_idx    (tile t + 2): slot of the lane's MFMA row, its edge id, owner and gathered node
//   wm_row    (tile t + 1): segment bounds of the owner -> FIRST/LAST flags; the record (A operand); Y_lm
//   wm_table  (tile t + 1): publish the row's entries in the (double-buffered) tile table
struct WmIdx {
  int slot, p;
  bool valid;
  int eid, own, g;
};
__device__ __forceinline__ WmIdx wm_idx(const WmArgs& a, int lane, int t, int e0, int e1, int e2) {
  const int i = lane & 31, hr = (i >> 2) & 1, v = 4 * (i >> 3) + (i & 3);
  const int beg = hr ? e1 : e0, end = hr ? e2 : e1;
  WmIdx x;
  x.slot = beg + 16 * t + v;
  x.p = 16 * hr + v;
  x.valid = x.slot < end;
  const int sl = x.valid ? x.slot : e2 - 1;   // e2 > e0 whenever a tile exists
  x.eid = a.slot_eid ? a.slot_eid[sl] : sl;
  x.own = a.slot_owner[sl];
  x.g = a.slot_gather[sl];
  return x;
}
template <int KS, int NREC, bool WITH_Y>
struct WmRow {
  float R[NREC][KS];
  int flags;
  f32x4 ya, yb;
};
template <int KS, int NREC, bool WITH_Y>
__device__ __forceinline__ void wm_row(const WmArgs& a, const WmIdx& x, int kh, const float* __restrict__ rec,
                                       const float* __restrict__ drec, WmRow<KS, NREC, WITH_Y>& w) {
  constexpr int KP = (KS + 3) & ~3, EW = 2 * KP + 8;
  const int r_lo = a.rowptr[x.own], r_hi = a.rowptr[x.own + 1];
  const float* rp = rec + (int64_t)x.eid * EW;
  wm_load_record<KS>(rp, kh, x.valid, w.R[0]);
  if constexpr (NREC > 1) wm_load_record<KS>(drec + (int64_t)x.eid * EW, kh, x.valid, w.R[1]);
  if constexpr (WITH_Y) {
    w.ya = *reinterpret_cast<const f32x4*>(rp + 2 * KP);
    w.yb = *reinterpret_cast<const f32x4*>(rp + 2 * KP + 4);
  }
  int flags = x.valid ? WM_VALID : 0;
  if (x.valid && x.slot == r_lo) flags |= WM_FIRST;
  if (x.valid && x.slot + 1 == r_hi) flags |= WM_LAST;
  w.flags = flags;
}
template <int KS, int NREC, bool WITH_Y>
__device__ __forceinline__ void wm_table(int lane, const WmIdx& x, const WmRow<KS, NREC, WITH_Y>& w, int* tbl) {
  if (lane < 32) {
    tbl[T_GOFF * 32 + x.p] = x.valid ? x.g : 0;
    tbl[T_OWN * 32 + x.p] = x.own;
    tbl[T_FLAGS * 32 + x.p] = w.flags;
    tbl[T_EID * 32 + x.p] = x.eid;
    if constexpr (WITH_Y) {
      float* tf = reinterpret_cast<float*>(tbl);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tf[(T_Y + q) * 32 + x.p] = w.ya[q];
        tf[(T_Y + 4 + q) * 32 + x.p] = w.yb[q];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward
// Common prologue of a wave: its two streams and the columns of its unit.
struct WmWave {
  int e0, e1, e2, ntiles;
  int col_hs, col_he, col_hm, col_s, col_xe;   // per-lane columns (floats) in rows of h | s | e3nn x
  int64_t col_x;                               // per-lane offset in xhat (+ g * xnode + m * xcomp)
  int64_t xnode;
  int xcomp;
};
template <int NM>
__device__ __forceinline__ WmWave wm_wave(const WmArgs& a, int range, const WmUnit& un, int j) {
  WmWave w;
  const int n0 = a.stream_ptr[2 * range], n1 = a.stream_ptr[2 * range + 1], n2 = a.stream_ptr[2 * range + 2];
  w.e0 = a.rowptr[n0];
  w.e1 = a.rowptr[n1];
  w.e2 = a.rowptr[n2];
  const int len0 = w.e1 - w.e0, len1 = w.e2 - w.e1;
  w.ntiles = ((len0 > len1 ? len0 : len1) + 15) >> 4;
  const XAddr xa = xaddr(a.ir, a.n_nodes, un.u0, a.xl);
  w.col_hs = un.u0 + j;
  w.col_he = a.C + un.u0 + j;
  w.col_hm = 2 * a.C + 32 * un.cb + j;
  w.col_s = 32 * un.cb + j;
  w.col_xe = un.xbase + j * NM;
  w.col_x = xa.off + (int64_t)j * (a.xl == 0 ? NM : 1);
  w.xnode = xa.node;
  w.xcomp = xa.comp;
  return w;
}

// One role of the forward pass.  Each tile is walked in passes so that at most two 32x32 accumulators are live:
//   pass X (phi_state, phi_edge):  x_c += xhat[n] (h_state[n] phi_state) + Y (h_edge[n] phi_edge)
//   pass M (phi_msg, l = 0 only):  s_c += h_msg[n] phi_msg
// Every pass keeps its own running sums across rows and tiles and stores them when its segment ends.
template <int NM, int KS>
__device__ __forceinline__ void wm_fwd_body(const WmArgs& a, int range, const WmUnit un, const float* __restrict__ rec,
                                            const float* __restrict__ h, const float* __restrict__ xhat,
                                            const float* __restrict__ s_in, const float* __restrict__ x_in,
                                            const float* __restrict__ w_rbf, const float* __restrict__ b_rbf,
                                            float* __restrict__ s_out, float* __restrict__ x_out, int* tbl) {
  constexpr bool HAS_S = NM == 1;
  constexpr int YOFF = NM == 3 ? 0 : 3;   // Y1 at table rows T_Y + 0..2, Y2 at T_Y + 3..7
  constexpr int CH = NM == 5 ? 2 : 4;     // rows per gather chunk (double-buffered)
  const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int B = a.B, C = a.C, H = a.H, D = a.D, F = a.F;
  const WmWave wv = wm_wave<NM>(a, range, un, j);
  if (wv.ntiles == 0) return;
  const int e0 = wv.e0, e1 = wv.e1, e2 = wv.e2;

  float Ws[KS], We[KS], Wm[HAS_S ? KS : 1];
  wm_load_weights<KS>(w_rbf, b_rbf, un.u0 + j, B, hh, Ws);
  wm_load_weights<KS>(w_rbf, b_rbf, C + un.u0 + j, B, hh, We);
  if constexpr (HAS_S) wm_load_weights<KS>(w_rbf, b_rbf, 2 * C + 32 * un.cb + j, B, hh, Wm);

  float acc_s = 0.f, res_s = 0.f, acc_x[NM], res_x[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) acc_x[m] = res_x[m] = 0.f;

  struct ChunkX {
    float hs[CH], he[CH], xv[CH][NM];
  };
  auto load_x = [&](const int* trow, int c0, ChunkX& c) {
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const int g = trow[T_GOFF * 32 + c0 + r];
      const float* hn = h + (int64_t)g * H;
      c.hs[r] = hn[wv.col_hs];
      c.he[r] = hn[wv.col_he];
      const float* xn = xhat + wv.col_x + (int64_t)g * wv.xnode;
#pragma unroll
      for (int m = 0; m < NM; ++m) c.xv[r][m] = xn[m * wv.xcomp];
    }
  };

  using Row = WmRow<KS, 1, (NM > 1)>;
  WmIdx ix = wm_idx(a, lane, 0, e0, e1, e2);
  Row row;
  wm_row<KS, 1, (NM > 1)>(a, ix, hh, rec, nullptr, row);
  wm_table<KS, 1, (NM > 1)>(lane, ix, row, tbl);
  ix = wm_idx(a, lane, 1, e0, e1, e2);
  __builtin_amdgcn_wave_barrier();

  for (int t = 0; t < wv.ntiles; ++t) {
    const int* trow = tbl + (t & 1) * (T_ROWS * 32) + 16 * hh;   // + quantity * 32 + v
    int* tnext = tbl + ((t + 1) & 1) * (T_ROWS * 32);
    // ---- loads that fly under the MFMAs: first gather chunk, the msg rows, next tile's record, indices of t + 2
    ChunkX cx[2];
    load_x(trow, 0, cx[0]);
    float hm[HAS_S ? 16 : 1];
    if constexpr (HAS_S) {
#pragma unroll
      for (int v = 0; v < 16; ++v) hm[v] = h[(int64_t)trow[T_GOFF * 32 + v] * H + wv.col_hm];
    }
    float R[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) R[s] = row.R[0][s];
    const WmIdx ixn = ix;
    wm_row<KS, 1, (NM > 1)>(a, ixn, hh, rec, nullptr, row);
    ix = wm_idx(a, lane, t + 2, e0, e1, e2);
    __builtin_amdgcn_sched_barrier(0);
    {  // ---- pass X
      const f32x16 ds = wm_filter<KS>(R, Ws), de = wm_filter<KS>(R, We);
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += CH) {
        if (c0 + CH < 16) load_x(trow, c0 + CH, cx[((c0 / CH) + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const ChunkX& c = cx[(c0 / CH) & 1];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
          const int v = c0 + r;
          const int flags = trow[T_FLAGS * 32 + v];
          if (flags & WM_FIRST) {   // half-uniform: a new segment starts with this row
            const int64_t own = trow[T_OWN * 32 + v];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              res_x[m] = x_in[own * D + wv.col_xe + m];
              acc_x[m] = 0.f;
            }
          }
          const float gs = c.hs[r] * ds[v], ge = c.he[r] * de[v];
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            const float y = NM > 1 ? reinterpret_cast<const float*>(trow)[(T_Y + YOFF + m) * 32 + v] : 1.f;
            acc_x[m] += c.xv[r][m] * gs + y * ge;
          }
          if (flags & WM_LAST) {    // the segment ends with this row: its only store
            const int64_t own = trow[T_OWN * 32 + v];
#pragma unroll
            for (int m = 0; m < NM; ++m) x_out[own * D + wv.col_xe + m] = res_x[m] + acc_x[m];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (HAS_S) {  // ---- pass M
      const f32x16 dm = wm_filter<KS>(R, Wm);
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int flags = trow[T_FLAGS * 32 + v];
        if (flags & WM_FIRST) {
          const int64_t own = trow[T_OWN * 32 + v];
          res_s = s_in[own * F + wv.col_s];
          acc_s = 0.f;
        }
        acc_s += hm[v] * dm[v];
        if (flags & WM_LAST) {
          const int64_t own = trow[T_OWN * 32 + v];
          s_out[own * F + wv.col_s] = res_s + acc_s;
        }
      }
    }
    wm_table<KS, 1, (NM > 1)>(lane, ixn, row, tnext);
    __builtin_amdgcn_wave_barrier();
  }
}

// one kernel per l: each role gets its own register allocation
template <int NM, int KS>
__global__ void __launch_bounds__(64 * WM_WAVES) k_message_fwd_wm(WmArgs a, int unit0, int nunits, const float* __restrict__ rec,
                                                                  const float* __restrict__ h, const float* __restrict__ xhat,
                                                                  const float* __restrict__ s_in, const float* __restrict__ x_in,
                                                                  const float* __restrict__ w_rbf, const float* __restrict__ b_rbf,
                                                                  float* __restrict__ s_out, float* __restrict__ x_out) {
  __shared__ int tbl_all[WM_WAVES][2 * T_ROWS * 32];   // double-buffered tile table per wave
  int range, unit;
  if (!wm_decode(a, nunits, range, unit)) return;
  const WmUnit un = wm_unit(a, unit0 + unit);
  wm_fwd_body<NM, KS>(a, range, un, rec, h, xhat, s_in, x_in, w_rbf, b_rbf, s_out, x_out, tbl_all[threadIdx.x >> 6]);
}

// Nodes without edges are never touched by a stream: forward s_out = s_in, x_out = x_in; reverse grad_h = 0,
// grad_xhat = 0.  One lane per node finds them; the wave then writes each such row together.
__global__ void k_wm_isolated_fwd(int64_t N, const int32_t* __restrict__ rowptr, int F, int D, const float* __restrict__ s_in,
                                  const float* __restrict__ x_in, float* __restrict__ s_out, float* __restrict__ x_out) {
  const int lane = threadIdx.x & 63;
  const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
  const int64_t n = base + lane;
  unsigned long long mask = __ballot(n < N && rowptr[n] == rowptr[n + 1]);
  while (mask) {
    const int64_t m = base + (__ffsll((long long)mask) - 1);
    mask &= mask - 1;
    for (int f = lane; f < F; f += 64) s_out[m * F + f] = s_in[m * F + f];
    for (int f = lane; f < D; f += 64) x_out[m * D + f] = x_in[m * D + f];
  }
}
__global__ void k_wm_isolated_bwd(int64_t N, const int32_t* __restrict__ rowptr, int H, Irreps ir, int xl,
                                  float* __restrict__ grad_h, float* __restrict__ grad_xhat) {
  const int lane = threadIdx.x & 63;
  const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
  const int64_t n = base + lane;
  unsigned long long mask = __ballot(n < N && rowptr[n] == rowptr[n + 1]);
  const int C = ir.C();
  while (mask) {
    const int64_t m = base + (__ffsll((long long)mask) - 1);
    mask &= mask - 1;
    for (int f = lane; f < H; f += 64) grad_h[m * H + f] = 0.f;
    for (int u = lane; u < C; u += 64) {
      const XAddr xa = xaddr(ir, N, u, xl);
      int l, off;
      ir.locate(u, l, off);
      for (int c = 0; c < 2 * l + 1; ++c) grad_xhat[xa.off + m * xa.node + (int64_t)c * xa.comp] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------ reverse
struct WmParts {
  float* pd;   // [NU][E]      per-unit partial of dL/dd
  float* y1;   // [nu1][3][E]  per-unit partial of dL/dY_1m
  float* y2;   // [nu2][5][E]
};

// sum over the 32 lanes of each half-wave; the result is valid in lanes 16..31 (half 0) and 48..63 (half 1)
__device__ __forceinline__ float wm_half_total(float v) {
#define XEQ_WM_DPP(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xF, true)))
  v = XEQ_WM_DPP(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  v = XEQ_WM_DPP(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  v = XEQ_WM_DPP(v, 0x141, 0xF);   // row_half_mirror
  v = XEQ_WM_DPP(v, 0x140, 0xF);   // row_mirror
  v = XEQ_WM_DPP(v, 0x142, 0xA);   // row_bcast15 into rows 1 and 3
#undef XEQ_WM_DPP
  return v;
}

// One role of the reverse pass, in passes of two accumulators (value and d/dd filter of one kind):
//   pass S (state): g_hs[n] += phi_s dgs, g_xhat[n] += h_s[n] phi_s gx[c];  pd += h_s[n] dgs phi_s',  dgs = <xhat[n], gx[c]>
//   pass E (edge):  g_he[n] += phi_e dge;  pd += h_e[n] dge phi_e';  dL/dY_m = sum_ch h_e[n] phi_e gx[c][m],  dge = <Y, gx[c]>
//   pass M (msg):   g_hm[n] += phi_m gs[c]; pd += h_m[n] gs[c] phi_m'
// pd (dL/dd of the row's edge, this lane's channel) is carried across the passes in 16 registers.
template <int NM, int KS>
__device__ __forceinline__ void wm_bwd_body(const WmArgs& a, int range, int unit, const WmUnit un, const float* __restrict__ rec,
                                            const float* __restrict__ drec, const float* __restrict__ h,
                                            const float* __restrict__ xhat, const float* __restrict__ grad_s,
                                            const float* __restrict__ grad_x, const float* __restrict__ w_rbf,
                                            const float* __restrict__ b_rbf, float* __restrict__ grad_h,
                                            float* __restrict__ grad_xhat, const WmParts parts, int* tbl) {
  constexpr bool HAS_S = NM == 1;
  constexpr int YOFF = NM == 3 ? 0 : 3;
  constexpr int CH = NM == 5 ? 2 : 4;
  const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int B = a.B, C = a.C, H = a.H, D = a.D, F = a.F;
  const int64_t E = a.n_edges;
  const WmWave wv = wm_wave<NM>(a, range, un, j);
  if (wv.ntiles == 0) return;
  const int e0 = wv.e0, e1 = wv.e1, e2 = wv.e2;

  float Ws[KS], We[KS], Wm[HAS_S ? KS : 1];
  wm_load_weights<KS>(w_rbf, b_rbf, un.u0 + j, B, hh, Ws);
  wm_load_weights<KS>(w_rbf, b_rbf, C + un.u0 + j, B, hh, We);
  if constexpr (HAS_S) wm_load_weights<KS>(w_rbf, b_rbf, 2 * C + 32 * un.cb + j, B, hh, Wm);

  // the owner (neighbor) node of the running segment, per pass: its own rows and its gradient sums
  float o_hs = 0.f, o_he = 0.f, o_hm = 0.f, o_x[NM];
  float a_hs = 0.f, a_he = 0.f, a_hm = 0.f, a_x[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) o_x[m] = a_x[m] = 0.f;
  const bool writer = j == 31;   // lanes 31 / 63 hold the half's DPP totals

  struct ChunkG {   // gradient rows of the centers of CH edges
    float gx[CH][NM];
  };
  auto load_g = [&](const int* trow, int c0, ChunkG& c) {
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const float* gr = grad_x + (int64_t)trow[T_GOFF * 32 + c0 + r] * D + wv.col_xe;
#pragma unroll
      for (int m = 0; m < NM; ++m) c.gx[r][m] = gr[m];
    }
  };

  using Row = WmRow<KS, 2, (NM > 1)>;
  WmIdx ix = wm_idx(a, lane, 0, e0, e1, e2);
  Row row;
  wm_row<KS, 2, (NM > 1)>(a, ix, hh, rec, drec, row);
  wm_table<KS, 2, (NM > 1)>(lane, ix, row, tbl);
  ix = wm_idx(a, lane, 1, e0, e1, e2);
  __builtin_amdgcn_wave_barrier();

  for (int t = 0; t < wv.ntiles; ++t) {
    const int* trow = tbl + (t & 1) * (T_ROWS * 32) + 16 * hh;
    int* tnext = tbl + ((t + 1) & 1) * (T_ROWS * 32);
    ChunkG cg[2];
    load_g(trow, 0, cg[0]);
    float R[KS], Rd[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      R[s] = row.R[0][s];
      Rd[s] = row.R[1][s];
    }
    const WmIdx ixn = ix;
    ix = wm_idx(a, lane, t + 2, e0, e1, e2);
    __builtin_amdgcn_sched_barrier(0);
    float pd[16];
    {  // ---- pass S
      const f32x16 ds = wm_filter<KS>(R, Ws), qs = wm_filter<KS>(Rd, Ws);
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += CH) {
        if (c0 + CH < 16) load_g(trow, c0 + CH, cg[((c0 / CH) + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const ChunkG& c = cg[(c0 / CH) & 1];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
          const int v = c0 + r;
          const int flags = trow[T_FLAGS * 32 + v];
          if (flags & WM_FIRST) {   // new owner: load its rows, clear its sums
            const int64_t own = trow[T_OWN * 32 + v];
            o_hs = h[own * H + wv.col_hs];
            const float* xn = xhat + wv.col_x + own * wv.xnode;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              o_x[m] = xn[m * wv.xcomp];
              a_x[m] = 0.f;
            }
            a_hs = 0.f;
          }
          float dgs = 0.f;
#pragma unroll
          for (int m = 0; m < NM; ++m) dgs += o_x[m] * c.gx[r][m];
          const float ps = ds[v];
          a_hs += ps * dgs;
          const float gate = o_hs * ps;
#pragma unroll
          for (int m = 0; m < NM; ++m) a_x[m] += gate * c.gx[r][m];
          pd[v] = o_hs * dgs * qs[v];
          if (flags & WM_LAST) {
            const int64_t own = trow[T_OWN * 32 + v];
            grad_h[own * H + wv.col_hs] = a_hs;
            float* gxh = grad_xhat + wv.col_x + own * wv.xnode;
#pragma unroll
            for (int m = 0; m < NM; ++m) gxh[m * wv.xcomp] = a_x[m];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // next tile's records fly under the remaining passes
    wm_row<KS, 2, (NM > 1)>(a, ixn, hh, rec, drec, row);
    load_g(trow, 0, cg[0]);
    float gsv[HAS_S ? 16 : 1];
    if constexpr (HAS_S) {
#pragma unroll
      for (int v = 0; v < 16; ++v) gsv[v] = grad_s[(int64_t)trow[T_GOFF * 32 + v] * F + wv.col_s];
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // ---- pass E
      const f32x16 de = wm_filter<KS>(R, We), qe = wm_filter<KS>(Rd, We);
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += CH) {
        if (c0 + CH < 16) load_g(trow, c0 + CH, cg[((c0 / CH) + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const ChunkG& c = cg[(c0 / CH) & 1];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
          const int v = c0 + r;
          const int flags = trow[T_FLAGS * 32 + v];
          if (flags & WM_FIRST) {
            const int64_t own = trow[T_OWN * 32 + v];
            o_he = h[own * H + wv.col_he];
            a_he = 0.f;
          }
          float dge = 0.f;
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            const float y = NM > 1 ? reinterpret_cast<const float*>(trow)[(T_Y + YOFF + m) * 32 + v] : 1.f;
            dge += y * c.gx[r][m];
          }
          const float pe = de[v];
          a_he += pe * dge;
          pd[v] += o_he * dge * qe[v];
          if constexpr (NM > 1) {   // dL/dY_lm of the row's edge: sum over the unit's 32 channels
            const float wy = o_he * pe;
            float ry[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) ry[m] = wm_half_total(wy * c.gx[r][m]);
            if (writer && (flags & WM_VALID)) {
              const int eid = trow[T_EID * 32 + v];
              float* dst = NM == 3 ? parts.y1 : parts.y2;
#pragma unroll
              for (int m = 0; m < NM; ++m) dst[((int64_t)un.cb * NM + m) * E + eid] = ry[m];
            }
          }
          if (flags & WM_LAST) {
            const int64_t own = trow[T_OWN * 32 + v];
            grad_h[own * H + wv.col_he] = a_he;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (HAS_S) {  // ---- pass M
      const f32x16 dm = wm_filter<KS>(R, Wm), qm = wm_filter<KS>(Rd, Wm);
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int flags = trow[T_FLAGS * 32 + v];
        if (flags & WM_FIRST) {
          const int64_t own = trow[T_OWN * 32 + v];
          o_hm = h[own * H + wv.col_hm];
          a_hm = 0.f;
        }
        a_hm += dm[v] * gsv[v];
        pd[v] += o_hm * gsv[v] * qm[v];
        if (flags & WM_LAST) {
          const int64_t own = trow[T_OWN * 32 + v];
          grad_h[own * H + wv.col_hm] = a_hm;
        }
      }
    }
    // ---- dL/dd of every row's edge: sum over the unit's 32 channels
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float tot = wm_half_total(pd[v]);
      if (writer && (trow[T_FLAGS * 32 + v] & WM_VALID)) parts.pd[(int64_t)unit * E + trow[T_EID * 32 + v]] = tot;
    }
    wm_table<KS, 2, (NM > 1)>(lane, ixn, row, tnext);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int NM, int KS>
__global__ void __launch_bounds__(64 * WM_WAVES) k_message_bwd_wm(WmArgs a, int unit0, int nunits, const float* __restrict__ rec,
                                                                  const float* __restrict__ drec, const float* __restrict__ h,
                                                                  const float* __restrict__ xhat, const float* __restrict__ grad_s,
                                                                  const float* __restrict__ grad_x, const float* __restrict__ w_rbf,
                                                                  const float* __restrict__ b_rbf, float* __restrict__ grad_h,
                                                                  float* __restrict__ grad_xhat, WmParts parts) {
  __shared__ int tbl_all[WM_WAVES][2 * T_ROWS * 32];
  int range, unit;
  if (!wm_decode(a, nunits, range, unit)) return;
  const WmUnit un = wm_unit(a, unit0 + unit);
  wm_bwd_body<NM, KS>(a, range, unit0 + unit, un, rec, drec, h, xhat, grad_s, grad_x, w_rbf, b_rbf, grad_h, grad_xhat, parts,
                      tbl_all[threadIdx.x >> 6]);
}

// dL/dvec from the per-unit partials, summed in unit order (deterministic), chain rule of A1-A3
__global__ void k_wm_edge_grad(const float* __restrict__ vec, int64_t E, int nu, int nu1, int nu2, const float* __restrict__ pd,
                               const float* __restrict__ y1, const float* __restrict__ y2, float* __restrict__ grad_vec) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float gd = 0.f, q1[3] = {0.f, 0.f, 0.f}, q2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int u = 0; u < nu; ++u) gd += pd[(int64_t)u * E + e];
  for (int u = 0; u < nu1; ++u)
#pragma unroll
    for (int m = 0; m < 3; ++m) q1[m] += y1[((int64_t)u * 3 + m) * E + e];
  for (int u = 0; u < nu2; ++u)
#pragma unroll
    for (int m = 0; m < 5; ++m) q2[m] += y2[((int64_t)u * 5 + m) * E + e];
  const EdgeGeom<float> g = edge_geom<float>(vec[3 * e], vec[3 * e + 1], vec[3 * e + 2]);
  float out[3];
  edge_grad<float>(g, gd, q1, q2, out);
  grad_vec[3 * e] = out[0];
  grad_vec[3 * e + 1] = out[1];
  grad_vec[3 * e + 2] = out[2];
}

static bool wm_supported(int num_basis, int node_dim, const int32_t mul[3]) {
  return num_basis >= 1 && num_basis <= 31 && mul[0] == node_dim && mul[0] > 0 && mul[0] % 32 == 0 && mul[1] >= 0 &&
         mul[1] % 32 == 0 && mul[2] >= 0 && mul[2] % 32 == 0;
}

static int wm_check(const char* who, int64_t n_nodes, int64_t n_edges, int n_ranges, int num_basis, int node_dim,
                    const int32_t mul[3], WmArgs& a) {
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (1ll << 31) && n_nodes < (1ll << 31), "%s: bad sizes", who);
  XEQ_CHECK_ARG(n_ranges >= 0, "%s: bad range table", who);
  if (!wm_supported(num_basis, node_dim, mul)) {
    xeq::set_error("%s: the matrix-core form needs node_dim == mul[0], multiplicities in multiples of 32 and num_basis <= 31", who);
    return XEQ_ERR_UNSUPPORTED;
  }
  for (int l = 0; l < 3; ++l) a.ir.mul[l] = mul[l];
  a.C = a.ir.C();
  a.D = a.ir.D();
  a.F = node_dim;
  a.H = a.F + 2 * a.C;
  a.B = num_basis;
  XEQ_CHECK_ARG(n_nodes * (int64_t)a.H < (1ll << 31) && n_edges * (int64_t)72 < (1ll << 31),
                "%s: tensors too large for 32-bit offsets (shard the batch)", who);
  for (int l = 0; l < 3; ++l) a.nu[l] = mul[l] / 32;
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.n_ranges = n_ranges;
  return XEQ_OK;
}

}  // namespace xeq

using namespace xeq;

// KS covers K = B + 1 (bias column) in steps of two; records are written for the same bucket.  One launch per l.
#define XEQ_WM_LAUNCH_L(KERNEL, NM_, ...)                                                                                 \
  do {                                                                                                                    \
    const int ks = wm_ks(num_basis);                                                                                      \
    if (ks <= 5) hipLaunchKernelGGL((KERNEL<NM_, 5>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__);      \
    else if (ks <= 9) hipLaunchKernelGGL((KERNEL<NM_, 9>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (ks <= 11) hipLaunchKernelGGL((KERNEL<NM_, 11>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<NM_, 16>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__);             \
  } while (0)
#define XEQ_WM_DISPATCH(KERNEL, ...)                                                            \
  do {                                                                                          \
    int unit0 = 0;                                                                              \
    for (int l = 0; l < 3; ++l) {                                                               \
      const int nunits = a.nu[l];                                                               \
      if (nunits == 0) continue;                                                                \
      dim3 grid(wm_grid(n_ranges, nunits));                                                     \
      if (l == 0) XEQ_WM_LAUNCH_L(KERNEL, 1, a, unit0, nunits, __VA_ARGS__);                    \
      else if (l == 1) XEQ_WM_LAUNCH_L(KERNEL, 3, a, unit0, nunits, __VA_ARGS__);               \
      else XEQ_WM_LAUNCH_L(KERNEL, 5, a, unit0, nunits, __VA_ARGS__);                           \
      unit0 += nunits;                                                                          \
    }                                                                                           \
  } while (0)

// the template KS a given num_basis is dispatched to; the records are laid out for it
static int wm_ks_bucket(int num_basis) {
  const int ks = wm_ks(num_basis);
  return ks <= 5 ? 5 : (ks <= 9 ? 9 : (ks <= 11 ? 11 : 16));
}
static int wm_kp_bucket(int num_basis) { return (wm_ks_bucket(num_basis) + 3) & ~3; }
static unsigned wm_grid(int n_ranges, int nunits) {
  const int64_t items = (int64_t)n_ranges * nunits;
  int64_t blocks = (items + WM_WAVES - 1) / WM_WAVES;
  if (blocks >= 64) blocks = (blocks + 7) / 8 * 8;   // multiple of 8: XCD-aware item order (wm_decode)
  return (unsigned)blocks;
}

extern "C" {

int xeq_edge_basis_wm_width(int num_basis) { return 2 * wm_kp_bucket(num_basis) + 8; }

int xeq_edge_basis_wm(const void* vec, int64_t n_edges, int rbf_kind, int cutoff_kind, int num_basis, double cutoff,
                      const void* p0, const void* p1, void* basis, void* dbasis, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && num_basis >= 1 && num_basis <= 31 && cutoff > 0, "xeq_edge_basis_wm: bad sizes");
  XEQ_CHECK_ARG(rbf_kind == XEQ_RBF_BESSEL || rbf_kind == XEQ_RBF_GAUSSIAN, "xeq_edge_basis_wm: rbf kernel %d is not implemented", rbf_kind);
  XEQ_CHECK_ARG(rbf_kind != XEQ_RBF_GAUSSIAN || p1 != nullptr, "xeq_edge_basis_wm: gaussian rbf needs std");
  XEQ_CHECK_ARG(cutoff_kind == XEQ_CUTOFF_COSINE || cutoff_kind == XEQ_CUTOFF_POLYNOMIAL, "xeq_edge_basis_wm: cutoff function %d is not implemented", cutoff_kind);
  if (n_edges == 0) return XEQ_OK;
  RadialSpec rs{rbf_kind, cutoff_kind, num_basis, cutoff};
  const int KP = wm_kp_bucket(num_basis);
  const int64_t total = n_edges * (2 * KP + 8);
  XEQ_CHECK_ARG(total < (1ll << 31), "xeq_edge_basis_wm: too many edges for 32-bit record offsets (shard the batch)");
  hipLaunchKernelGGL(k_edge_basis_wm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)vec, n_edges, rs, KP, (const float*)p0, (const float*)p1, (float*)basis, (float*)dbasis);
  XEQ_CHECK_LAUNCH("xeq_edge_basis_wm");
  return XEQ_OK;
}

int xeq_message_wm_supported(int num_basis, int node_dim, const int32_t mul[3]) { return wm_supported(num_basis, node_dim, mul) ? 1 : 0; }

int xeq_message_fwd_wm(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* stream_ptr, const int32_t* c_rowptr,
                       const int32_t* slot_eid, const int32_t* slot_center, const int32_t* slot_nbr, const void* basis,
                       const void* h, const void* xhat, const void* s_in, const void* x_in, const void* w_rbf,
                       const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* s_out, void* x_out,
                       int xhat_layout, void* stream) {
  WmArgs a{};
  int rcode = wm_check("xeq_message_fwd_wm", n_nodes, n_edges, n_ranges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.stream_ptr = stream_ptr;
  a.rowptr = c_rowptr;
  a.slot_eid = slot_eid;
  a.slot_owner = slot_center;
  a.slot_gather = slot_nbr;
  a.xl = xhat_layout;
  hipLaunchKernelGGL(k_wm_isolated_fwd, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_nodes,
                     c_rowptr, a.F, a.D, (const float*)s_in, (const float*)x_in, (float*)s_out, (float*)x_out);
  XEQ_CHECK_LAUNCH("xeq_message_fwd_wm (isolated nodes)");
  if (n_ranges == 0 || n_edges == 0) return XEQ_OK;
  XEQ_WM_DISPATCH(k_message_fwd_wm, (const float*)basis, (const float*)h, (const float*)xhat, (const float*)s_in,
                  (const float*)x_in, (const float*)w_rbf, (const float*)b_rbf, (float*)s_out, (float*)x_out);
  XEQ_CHECK_LAUNCH("xeq_message_fwd_wm");
  return XEQ_OK;
}

int xeq_message_bwd_wm(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* stream_ptr, const int32_t* n_rowptr,
                       const int32_t* slot_eid, const int32_t* slot_nbr, const int32_t* slot_center, const void* basis,
                       const void* dbasis, const void* h, const void* xhat, const void* grad_s, const void* grad_x,
                       const void* w_rbf, const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* grad_h,
                       void* grad_xhat, void* parts, int xhat_layout, void* stream) {
  WmArgs a{};
  int rcode = wm_check("xeq_message_bwd_wm", n_nodes, n_edges, n_ranges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.stream_ptr = stream_ptr;
  a.rowptr = n_rowptr;
  a.slot_eid = slot_eid;
  a.slot_owner = slot_nbr;
  a.slot_gather = slot_center;
  a.xl = xhat_layout;
  hipLaunchKernelGGL(k_wm_isolated_bwd, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_nodes,
                     n_rowptr, a.H, a.ir, a.xl, (float*)grad_h, (float*)grad_xhat);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_wm (isolated nodes)");
  if (n_ranges == 0 || n_edges == 0) return XEQ_OK;
  const int nunits = a.nu[0] + a.nu[1] + a.nu[2];
  WmParts pr;
  pr.pd = (float*)parts;
  pr.y1 = pr.pd + (int64_t)nunits * n_edges;
  pr.y2 = pr.y1 + (int64_t)a.nu[1] * 3 * n_edges;
  XEQ_WM_DISPATCH(k_message_bwd_wm, (const float*)basis, (const float*)dbasis, (const float*)h, (const float*)xhat,
                  (const float*)grad_s, (const float*)grad_x, (const float*)w_rbf, (const float*)b_rbf, (float*)grad_h,
                  (float*)grad_xhat, pr);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_wm");
  return XEQ_OK;
}

/* floats of the `parts` scratch buffer of xeq_message_bwd_wm */
int64_t xeq_message_wm_parts_floats(int64_t n_edges, const int32_t mul[3]) {
  return n_edges * (int64_t)(mul[0] / 32 + mul[1] / 32 + mul[2] / 32 + 3 * (mul[1] / 32) + 5 * (mul[2] / 32));
}

int xeq_message_wm_edge_grad(const void* vec, int64_t n_edges, const int32_t mul[3], const void* parts, void* grad_vec,
                             void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && mul[0] % 32 == 0 && mul[1] % 32 == 0 && mul[2] % 32 == 0, "xeq_message_wm_edge_grad: bad sizes");
  if (n_edges == 0) return XEQ_OK;
  const int nu1 = mul[1] / 32, nu2 = mul[2] / 32, nunits = mul[0] / 32 + nu1 + nu2;
  const float* pd = (const float*)parts;
  const float* y1 = pd + (int64_t)nunits * n_edges;
  const float* y2 = y1 + (int64_t)nu1 * 3 * n_edges;
  hipLaunchKernelGGL(k_wm_edge_grad, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)vec, n_edges, nunits, nu1, nu2, pd, y1, y2, (float*)grad_vec);
  XEQ_CHECK_LAUNCH("xeq_message_wm_edge_grad");
  return XEQ_OK;
}

}  // extern "C"
