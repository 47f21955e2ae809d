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
  const int32_t* perm;         // [E] edge id per slot of the walk order (NULL: identity)
  const int64_t* owner;        // [E] per EDGE: node whose segment it belongs to (forward: center, reverse: neighbor)
  const int64_t* gather;       // [E] per EDGE: node whose rows are gathered    (forward: neighbor, reverse: center)
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

// rows of a tile whose gathers are in flight together (register budget per role)
#ifndef XEQ_WM_FWD_GR
#define XEQ_WM_FWD_GR(NM) ((NM) == 1 ? 16 : ((NM) == 3 ? 8 : 4))
#endif
#ifndef XEQ_WM_BWD_GR
#define XEQ_WM_BWD_GR(NM) ((NM) == 5 ? 8 : 16)
#endif
// resident waves per SIMD the register allocation must allow
// scheduling fences of the tile loops (dev switch: -D'XEQ_WM_SB()=' compiles them out)
#ifndef XEQ_WM_SB
#define XEQ_WM_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef XEQ_WM_FWD_WPE
#define XEQ_WM_FWD_WPE 3
#endif
#ifndef XEQ_WM_BWD_WPE
#define XEQ_WM_BWD_WPE 2
#endif
constexpr int WM_WAVES = 4;   // waves per workgroup (independent of one another)

// (range, unit) of this wave.  The WM_WAVES waves of a workgroup work on the same unit (they share its rbf_lin rows,
// staged once in LDS) and on WM_WAVES consecutive ranges.  Consecutive work items are the units of one group of
// ranges (they share its edge records and index arrays) and are dealt to workgroups so that neighbours in that order
// run on the same XCD (blocks b and b + 8 share one under round-robin dispatch: speed only).
__device__ __forceinline__ void wm_decode(const WmArgs& a, int nunits, int& range, int& unit) {
  const int nb = gridDim.x, b = blockIdx.x;
  const int item = (nb & 7) == 0 ? (b & 7) * (nb >> 3) + (b >> 3) : b;
#ifdef XEQ_WM_UNIT_MAJOR   // all workgroups of one unit are adjacent in the grid: one role's code in flight at a time
  const int ngroups = (a.n_ranges + WM_WAVES - 1) / WM_WAVES;
  unit = item / ngroups;
  const int rgroup = unit < nunits ? item - unit * ngroups : ngroups;
  if (unit >= nunits) unit = 0;
#else
  const int rgroup = item / nunits;   // padding blocks of the grid land beyond the last group: all their ranges are empty
  unit = item - rgroup * nunits;
#endif
  range = rgroup * WM_WAVES + (threadIdx.x >> 6);   // may be >= n_ranges: such a wave only helps staging
}

// rbf_lin rows of the unit as the B operand, staged in LDS once per workgroup: wl[kind][s][lane] with lane
// (j = lane & 31 -> channel row0 + j, kh = lane >> 5) holding W~[row][2 s + kh], s < KS, W~[., B] = bias, zeros beyond.
// kind 0: gate_state rows, 1: gate_edge rows, 2: scalar-message rows (l = 0 units only).
template <int KS>
__device__ __forceinline__ void wm_stage_weights(const WmArgs& a, const WmUnit& un, const float* __restrict__ w,
                                                 const float* __restrict__ b, float* wl) {
  const int nkind = un.l == 0 ? 3 : 2, B = a.B;
  for (int idx = threadIdx.x; idx < nkind * KS * 64; idx += blockDim.x) {
    const int kind = idx / (KS * 64), rem = idx - kind * (KS * 64), sstep = rem >> 6, ln = rem & 63;
    const int row = (kind == 0 ? un.u0 : (kind == 1 ? a.C + un.u0 : 2 * a.C + 32 * un.cb)) + (ln & 31);
    const int k = 2 * sstep + (ln >> 5);
    wl[idx] = k < B ? w[(int64_t)row * B + k] : (k == B ? b[row] : 0.f);
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

// development switches (scratch/build_variant.sh -DXEQ_WM_ABLATE=bits): 1 no MFMA, 2 no gathers, 4 no channel sums
#ifndef XEQ_WM_ABLATE
#define XEQ_WM_ABLATE 0
#endif

// W: this lane's column of one kind in the staged weights (wl + kind * KS * 64 + lane)
template <int KS>
__device__ __forceinline__ f32x16 wm_filter(const float (&R)[KS], const float* W) {
  f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (XEQ_WM_ABLATE & 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = R[i % KS] * W[0];
    return d;
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(R[s], W[s * 64], d, 0, 0, 0);
  return d;
}

// The tile table (LDS, private to the wave, double-buffered): per position p = 16 * half + v (v-th edge of the
// half-wave's stream in this tile) what every lane of the half needs for that row, written by the lane that owns
// MFMA row i(p) = 8 (v >> 2) + 4 half + (v & 3) and read back with half-uniform addresses (broadcast reads).
// Row offsets are BYTES (32-bit: wm_check bounds the tensors), so a gather is one v_add and a saddr load.
enum { T_G0 = 0, T_G1 = 1, T_OWN = 2, T_EID = 3, T_Y = 4, T_ROWS = 12 };

__device__ __forceinline__ float wm_ld(const float* __restrict__ base, uint32_t byte_off) {
  if (XEQ_WM_ABLATE & 2) return __uint_as_float(byte_off | 0x3f000000u);
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void wm_st(float* __restrict__ base, uint32_t byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// Tile setup runs ahead of the arithmetic, in steps whose loads are issued one tile apart:
//   wm_idx    (tile t + 2): slot of the lane's MFMA row, its edge id, owner and gathered node
//   wm_row    (tile t + 1): segment bounds of the owner -> FIRST/LAST bits; the record (A operand); Y_lm
//   wm_table  (tile t + 1): publish the row's entries in the tile table; the bits become wave-level masks
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
  x.eid = a.perm ? a.perm[sl] : sl;
  x.own = (int)a.owner[x.eid];
  x.g = (int)a.gather[x.eid];
  return x;
}
// bit i of a mask = MFMA row i = position (half (i >> 2) & 1, v = 4 (i >> 3) + (i & 3))
struct WmMasks {
  uint32_t first, last, valid;
};
template <int KS, int NREC, bool WITH_Y>
struct WmRow {
  float R[NREC][KS];
  bool first, last;
  f32x4 ya, yb;
};
template <int KS, int NREC, bool WITH_Y>
__device__ __forceinline__ void wm_row(const WmArgs& a, const WmIdx& x, int kh, const float* __restrict__ rec,
                                       const float* __restrict__ drec, WmRow<KS, NREC, WITH_Y>& w) {
  constexpr int KP = (KS + 3) & ~3, EW = 2 * KP + 8;
  const int r_lo = a.rowptr[x.own], r_hi = a.rowptr[x.own + 1];
  const float* rp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(rec) + (uint32_t)x.eid * (uint32_t)(EW * 4));
  wm_load_record<KS>(rp, kh, x.valid, w.R[0]);
  if constexpr (NREC > 1) {
    const float* dp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(drec) + (uint32_t)x.eid * (uint32_t)(EW * 4));
    wm_load_record<KS>(dp, kh, x.valid, w.R[1]);
  }
  if constexpr (WITH_Y) {
    w.ya = *reinterpret_cast<const f32x4*>(rp + 2 * KP);
    w.yb = *reinterpret_cast<const f32x4*>(rp + 2 * KP + 4);
  }
  w.first = x.valid && x.slot == r_lo;
  w.last = x.valid && x.slot + 1 == r_hi;
}
template <int KS, int NREC, bool WITH_Y>
__device__ __forceinline__ WmMasks wm_table(int lane, const WmIdx& x, const WmRow<KS, NREC, WITH_Y>& w, uint32_t stride0,
                                            uint32_t stride1, int* tbl) {
  if (lane < 32) {
    const uint32_t g = x.valid ? (uint32_t)x.g : 0u;
    tbl[T_G0 * 32 + x.p] = (int)(g * stride0);
    tbl[T_G1 * 32 + x.p] = (int)(g * stride1);
    tbl[T_OWN * 32 + x.p] = x.own;
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
  WmMasks m;
  m.first = (uint32_t)__ballot(w.first);
  m.last = (uint32_t)__ballot(w.last);
  m.valid = (uint32_t)__ballot(x.valid);
  return m;
}
// N consecutive table entries (positions v0 .. v0 + N - 1 of one quantity) with one LDS read
template <int N, typename T>
__device__ __forceinline__ void wm_tread(const int* trow, int q, int v0, T (&out)[N]) {
  static_assert(sizeof(T) == 4, "table entries are dwords");
  typedef T vecN __attribute__((ext_vector_type(N)));
  const vecN v = *reinterpret_cast<const vecN*>(reinterpret_cast<const T*>(trow) + q * 32 + v0);
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = v[i];
}
// row v of the D layout: bit of half 0; half 1 is 4 higher
__device__ __forceinline__ constexpr int wm_bit(int v) { return (v & 3) + 8 * (v >> 2); }

// ------------------------------------------------------------------------------------------------ forward
// Common prologue of a wave: its two streams and the per-lane byte columns of its unit.
struct WmWave {
  int e0, e1, e2, ntiles;
  uint32_t b_hs, b_hm, b_s, b_xe;   // byte offset of the lane's column in a row of h (state | msg), s, e3nn x
  uint32_t b_x;                     // ... in xhat, after the unit's uniform base: + g * xnode_b + m * xcomp_b
  uint32_t xnode_b, xcomp_b;
  int64_t x_base;                   // uniform element offset of the unit in xhat
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
  w.b_hs = 4u * (uint32_t)(un.u0 + j);
  w.b_hm = 4u * (uint32_t)(2 * a.C + 32 * un.cb + j);
  w.b_s = 4u * (uint32_t)(32 * un.cb + j);
  w.b_xe = 4u * (uint32_t)(un.xbase + j * NM);
  w.b_x = 4u * (uint32_t)(j * (a.xl == 0 ? NM : 1));
  w.xnode_b = 4u * (uint32_t)xa.node;
  w.xcomp_b = 4u * (uint32_t)xa.comp;
  w.x_base = xa.off;
  return w;
}

// Owner nodes of the wave's two streams that have no edge: no tile ever touches them.  The wave copies (forward)
// or clears (reverse) its unit's columns of their rows.  fn(node) is called by all 64 lanes.
template <typename F>
__device__ __forceinline__ void wm_for_isolated(const WmArgs& a, int range, int lane, F fn) {
  const int n0 = a.stream_ptr[2 * range], n2 = a.stream_ptr[2 * range + 2];
  for (int base = n0; base < n2; base += 64) {
    const int n = base + lane;
    unsigned long long mask = __ballot(n < n2 && a.rowptr[n] == a.rowptr[n + 1]);
    while (mask) {
      const int m = base + (__ffsll((long long)mask) - 1);
      mask &= mask - 1;
      fn(m);
    }
  }
}

// One role of the forward pass.  Each tile is walked in passes so that at most two 32x32 accumulators are live:
//   pass X (phi_state, phi_edge):  x_c += xhat[n] (h_state[n] phi_state) + Y (h_edge[n] phi_edge)
//   pass M (phi_msg, l = 0 only):  s_c += h_msg[n] phi_msg
// Every pass keeps its own running sums across rows and tiles and stores them when its segment ends.
template <int NM, int KS>
__device__ __forceinline__ void wm_fwd_body(const WmArgs& a, int range, const WmUnit un, const float* __restrict__ rec,
                                            const float* __restrict__ h, const float* __restrict__ xhat_,
                                            const float* __restrict__ s_in, const float* __restrict__ x_in,
                                            const float* wl, float* __restrict__ s_out, float* __restrict__ x_out, int* tbl) {
  constexpr bool HAS_S = NM == 1;
  constexpr int YOFF = NM == 3 ? 0 : 3;   // Y1 at table rows T_Y + 0..2, Y2 at T_Y + 3..7
  constexpr int GR = XEQ_WM_FWD_GR(NM);   // rows whose gathers are in flight together
  const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int C = a.C;
  const WmWave wv = wm_wave<NM>(a, range, un, j);
  const uint32_t row_s = 4u * (uint32_t)a.F, row_x = 4u * (uint32_t)a.D;
  wm_for_isolated(a, range, lane, [&](int m) {   // s_out = s_in, x_out = x_in on the unit's columns
    if (hh == 0) {
      if constexpr (HAS_S) wm_st(s_out, (uint32_t)m * row_s + wv.b_s, wm_ld(s_in, (uint32_t)m * row_s + wv.b_s));
#pragma unroll
      for (int mm = 0; mm < NM; ++mm)
        wm_st(x_out, (uint32_t)m * row_x + wv.b_xe + 4u * mm, wm_ld(x_in, (uint32_t)m * row_x + wv.b_xe + 4u * mm));
    }
  });
  if (wv.ntiles == 0) return;
  const int e0 = wv.e0, e1 = wv.e1, e2 = wv.e2;
  // one uniform base pointer per gathered quantity: a gather is then base[row offset + lane column], no per-load VALU
  const float* __restrict__ h_e = h + C;                                              // gate_edge rows, same lane column
  const float* __restrict__ h_m = h + (2 * C + 32 * un.cb - un.u0);                   // scalar-message rows
  const float* xhat_m[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) xhat_m[m] = xhat_ + wv.x_base + (int64_t)m * (wv.xcomp_b / 4);
  const uint32_t stride0 = 4u * (uint32_t)a.H, stride1 = wv.xnode_b;

  const float* Ws = wl + lane;
  const float* We = wl + KS * 64 + lane;
  const float* Wm = wl + 2 * KS * 64 + lane;

  float acc_s = 0.f, res_s = 0.f, acc_x[NM], res_x[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) acc_x[m] = res_x[m] = 0.f;

  using Row = WmRow<KS, 1, (NM > 1)>;
  WmIdx ix = wm_idx(a, lane, 0, e0, e1, e2);
  Row row;
  wm_row<KS, 1, (NM > 1)>(a, ix, hh, rec, nullptr, row);
  WmMasks mk = wm_table<KS, 1, (NM > 1)>(lane, ix, row, stride0, stride1, tbl);
  ix = wm_idx(a, lane, 1, e0, e1, e2);
  __builtin_amdgcn_wave_barrier();

  for (int t = 0; t < wv.ntiles; ++t) {
    const int* trow = tbl + (t & 1) * (T_ROWS * 32) + 16 * hh;   // + quantity * 32 + v
    int* tnext = tbl + ((t + 1) & 1) * (T_ROWS * 32);
    const uint32_t mfirst = mk.first >> (4 * hh), mlast = mk.last >> (4 * hh);   // this half's bits at wm_bit(v)
    // ---- the gathers of the first GR rows, the msg rows, the next tile's record and the indices of the tile after
    //      that are issued here and land under the MFMAs
    float hs[GR], he[GR], xv[GR][NM], hm[HAS_S ? 16 : 1];
    auto load_group = [&](int r0) {
#pragma unroll
      for (int v0 = 0; v0 < GR; v0 += 4) {
        int g0[4], g1[4];
        wm_tread<4>(trow, T_G0, r0 + v0, g0);
        wm_tread<4>(trow, T_G1, r0 + v0, g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t oh = (uint32_t)g0[r] + wv.b_hs;
          const uint32_t ox = (uint32_t)g1[r] + wv.b_x;
          hs[v0 + r] = wm_ld(h, oh);
          he[v0 + r] = wm_ld(h_e, oh);
#pragma unroll
          for (int m = 0; m < NM; ++m) xv[v0 + r][m] = wm_ld(xhat_m[m], ox);
        }
      }
    };
    load_group(0);
    if constexpr (HAS_S) {
#pragma unroll
      for (int v0 = 0; v0 < 16; v0 += 4) {
        int g0[4];
        wm_tread<4>(trow, T_G0, v0, g0);
#pragma unroll
        for (int r = 0; r < 4; ++r) hm[v0 + r] = wm_ld(h_m, (uint32_t)g0[r] + wv.b_hs);
      }
    }
    float R[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) R[s] = row.R[0][s];
    const WmIdx ixn = ix;
    wm_row<KS, 1, (NM > 1)>(a, ixn, hh, rec, nullptr, row);
    ix = wm_idx(a, lane, t + 2, e0, e1, e2);
    XEQ_WM_SB();
    {  // ---- pass X
      const f32x16 ds = wm_filter<KS>(R, Ws), de = wm_filter<KS>(R, We);
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += GR) {
        if (r0 > 0) {
          XEQ_WM_SB();
          load_group(r0);
        }
#pragma unroll
        for (int c0 = r0; c0 < r0 + GR; c0 += 4) {
          float Y[NM][4];
          if constexpr (NM > 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) wm_tread<4>(trow, T_Y + YOFF + m, c0, Y[m]);
          }
          const uint32_t any_first = (mk.first >> wm_bit(c0)) & 0xFFu, any_last = (mk.last >> wm_bit(c0)) & 0xFFu;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = c0 + r, gv = v - r0;
            if (any_first) {   // wave-uniform: some row of this chunk starts a segment in some half
              if ((mfirst >> wm_bit(v)) & 1u) {
                const uint32_t ob = (uint32_t)trow[T_OWN * 32 + v] * row_x + wv.b_xe;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                  res_x[m] = wm_ld(x_in, ob + 4u * m);
                  acc_x[m] = 0.f;
                }
              }
            }
            const float gs = hs[gv] * ds[v], ge = he[gv] * de[v];
#pragma unroll
            for (int m = 0; m < NM; ++m) acc_x[m] += xv[gv][m] * gs + (NM > 1 ? Y[m][r] : 1.f) * ge;
            if (any_last) {    // some row of this chunk ends a segment: its only store
              if ((mlast >> wm_bit(v)) & 1u) {
                const uint32_t ob = (uint32_t)trow[T_OWN * 32 + v] * row_x + wv.b_xe;
#pragma unroll
                for (int m = 0; m < NM; ++m) wm_st(x_out, ob + 4u * m, res_x[m] + acc_x[m]);
              }
            }
          }
        }
      }
    }
    if constexpr (HAS_S) {  // ---- pass M
      const f32x16 dm = wm_filter<KS>(R, Wm);
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += 4) {
        const uint32_t any_first = (mk.first >> wm_bit(c0)) & 0xFFu, any_last = (mk.last >> wm_bit(c0)) & 0xFFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = c0 + r;
          if (any_first) {
            if ((mfirst >> wm_bit(v)) & 1u) {
              res_s = wm_ld(s_in, (uint32_t)trow[T_OWN * 32 + v] * row_s + wv.b_s);
              acc_s = 0.f;
            }
          }
          acc_s += hm[v] * dm[v];
          if (any_last) {
            if ((mlast >> wm_bit(v)) & 1u) wm_st(s_out, (uint32_t)trow[T_OWN * 32 + v] * row_s + wv.b_s, res_s + acc_s);
          }
        }
      }
    }
    mk = wm_table<KS, 1, (NM > 1)>(lane, ixn, row, stride0, stride1, tnext);
    __builtin_amdgcn_wave_barrier();
  }
}

// all roles in one launch: units [0, nu0) are l = 0, then l = 1, then l = 2
template <int KS>
__global__ void __launch_bounds__(64 * WM_WAVES) __attribute__((amdgpu_waves_per_eu(XEQ_WM_FWD_WPE))) k_message_fwd_wm(WmArgs a, const float* __restrict__ rec,
                                                                  const float* __restrict__ h, const float* __restrict__ xhat,
                                                                  const float* __restrict__ s_in, const float* __restrict__ x_in,
                                                                  const float* __restrict__ w_rbf, const float* __restrict__ b_rbf,
                                                                  float* __restrict__ s_out, float* __restrict__ x_out) {
  __shared__ int tbl_all[WM_WAVES][2 * T_ROWS * 32];   // double-buffered tile table per wave
  __shared__ float wl[3 * KS * 64];                     // the unit's rbf_lin rows (B operand of every MFMA)
  int range, unit;
  wm_decode(a, a.nu[0] + a.nu[1] + a.nu[2], range, unit);
  const WmUnit un = wm_unit(a, unit);
  wm_stage_weights<KS>(a, un, w_rbf, b_rbf, wl);
  __syncthreads();
  if (range >= a.n_ranges) return;
  int* tbl = tbl_all[threadIdx.x >> 6];
#ifdef XEQ_WM_ONLY_L   // development: register budget of one role
  if (un.l == XEQ_WM_ONLY_L) wm_fwd_body<2 * XEQ_WM_ONLY_L + 1, KS>(a, range, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl);
#else
  if (un.l == 0) wm_fwd_body<1, KS>(a, range, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl);
  else if (un.l == 1) wm_fwd_body<3, KS>(a, range, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl);
  else wm_fwd_body<5, KS>(a, range, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl);
#endif
}

// ------------------------------------------------------------------------------------------------ reverse
// development (-DXEQ_WM_STAMPS): cycles of one wave per phase of the reverse tile loop, summed over all l = 0 waves
__device__ unsigned long long g_wm_stamps[8];
#ifdef XEQ_WM_STAMPS
#define WM_STAMP(i)                                                 \
  do {                                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    st_[i] += now_ - last_;                                         \
    last_ = now_;                                                   \
  } while (0)
#else
#define WM_STAMP(i) \
  do {              \
  } while (0)
#endif

struct WmParts {
  float* pd;   // [NU][E]      per-unit partial of dL/dd
  float* y1;   // [nu1][3][E]  per-unit partial of dL/dY_1m
  float* y2;   // [nu2][5][E]
};

// sum over the 32 lanes of each half-wave; the result is valid in lanes 16..31 (half 0) and 48..63 (half 1)
__device__ __forceinline__ float wm_half_total(float v) {
  if (XEQ_WM_ABLATE & 4) return v;
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
                                            const float* __restrict__ xhat_, const float* __restrict__ grad_s,
                                            const float* __restrict__ grad_x, const float* wl, float* __restrict__ grad_h,
                                            float* __restrict__ grad_xhat_, const WmParts parts, int* tbl, float* ysc) {
  constexpr bool HAS_S = NM == 1;
  constexpr int YOFF = NM == 3 ? 0 : 3;
  constexpr int GR = XEQ_WM_BWD_GR(NM);   // rows whose gathers are in flight together
  const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int C = a.C;
  const int64_t E = a.n_edges;
  const WmWave wv = wm_wave<NM>(a, range, un, j);
  const float* __restrict__ xhat = xhat_ + wv.x_base;
  float* __restrict__ grad_xhat = grad_xhat_ + wv.x_base;
  const uint32_t he_off = 4u * (uint32_t)C, row_h = 4u * (uint32_t)a.H;
  wm_for_isolated(a, range, lane, [&](int m) {   // nobody's neighbor: zero gradients on the unit's columns
    if (hh == 0) {
      wm_st(grad_h, (uint32_t)m * row_h + wv.b_hs, 0.f);
      wm_st(grad_h, (uint32_t)m * row_h + wv.b_hs + he_off, 0.f);
      if constexpr (HAS_S) wm_st(grad_h, (uint32_t)m * row_h + wv.b_hm, 0.f);
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) wm_st(grad_xhat, (uint32_t)m * wv.xnode_b + wv.b_x + mm * wv.xcomp_b, 0.f);
    }
  });
  if (wv.ntiles == 0) return;
  const int e0 = wv.e0, e1 = wv.e1, e2 = wv.e2;
  const uint32_t stride0 = 4u * (uint32_t)a.D, stride1 = 4u * (uint32_t)a.F;   // gathered rows: grad_x, grad_s of the center

  const float* Ws = wl + lane;
  const float* We = wl + KS * 64 + lane;
  const float* Wm = wl + 2 * KS * 64 + lane;

  // the owner (neighbor) node of the running segment, per pass: its own rows and its gradient sums
  float o_hs = 0.f, o_he = 0.f, o_hm = 0.f, o_x[NM];
  float a_hs = 0.f, a_he = 0.f, a_hm = 0.f, a_x[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) o_x[m] = a_x[m] = 0.f;
  // wm_half_total leaves a half's total in its lanes 16..31: lane 16 + v keeps the total of row v, so the 16 rows of a
  // tile go out as ONE 64-byte store per quantity, indexed by the row's SLOT of the walk order (consecutive rows are
  // consecutive slots; xeq_message_wm_edge_grad maps slots back to edges).  Single-lane stores per row at the edge id
  // were 4-byte partial-line writes: 236 MB of WRITE_SIZE per launch against 101 MB of results.
  const int keep_v = j - 16;                                  // the row this lane keeps (lanes 16..31 of each half)
  const bool writer = j == 31;                                // one lane per half parks the dL/dY totals in LDS
  const int half_beg = hh ? e1 : e0, half_end = hh ? e2 : e1;

  using Row = WmRow<KS, 2, (NM > 1)>;
  WmIdx ix = wm_idx(a, lane, 0, e0, e1, e2);
  Row row;
  wm_row<KS, 2, (NM > 1)>(a, ix, hh, rec, drec, row);
  WmMasks mk = wm_table<KS, 2, (NM > 1)>(lane, ix, row, stride0, stride1, tbl);
  ix = wm_idx(a, lane, 1, e0, e1, e2);
  __builtin_amdgcn_wave_barrier();

#ifdef XEQ_WM_STAMPS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < wv.ntiles; ++t) {
    const int* trow = tbl + (t & 1) * (T_ROWS * 32) + 16 * hh;
    int* tnext = tbl + ((t + 1) & 1) * (T_ROWS * 32);
    const uint32_t mfirst = mk.first >> (4 * hh), mlast = mk.last >> (4 * hh), mvalid = mk.valid >> (4 * hh);
    (void)mvalid;
    // ---- the rows of the owners whose segments START in this tile are needed by the first row of the segment: the
    //      first two starts of each half are fetched here, under the MFMAs (a third start in one tile -- segments
    //      shorter than 8 edges -- takes the blocking path below)
    const uint32_t mf = mfirst & 0x0F0F0F0Fu;              // this half's rows only
    const uint32_t mf2 = mf & (mf - 1u);
    const int bA = mf ? __ffs((int)mf) - 1 : 0, bB = mf2 ? __ffs((int)mf2) - 1 : 0;
    const int vA = (bA & 3) + 4 * (bA >> 3), vB = mf2 ? (bB & 3) + 4 * (bB >> 3) : -1;
    float pA_hs, pA_he, pA_hm = 0.f, pA_x[NM], pB_hs, pB_he, pB_hm = 0.f, pB_x[NM];
    {
      const uint32_t ownA = (uint32_t)trow[T_OWN * 32 + vA], ownB = (uint32_t)trow[T_OWN * 32 + (vB < 0 ? vA : vB)];
      const uint32_t ohA = ownA * row_h + wv.b_hs, ohB = ownB * row_h + wv.b_hs;
      pA_hs = wm_ld(h, ohA);
      pA_he = wm_ld(h, ohA + he_off);
      pB_hs = wm_ld(h, ohB);
      pB_he = wm_ld(h, ohB + he_off);
      if constexpr (HAS_S) {
        pA_hm = wm_ld(h, ownA * row_h + wv.b_hm);
        pB_hm = wm_ld(h, ownB * row_h + wv.b_hm);
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        pA_x[m] = wm_ld(xhat, ownA * wv.xnode_b + wv.b_x + m * wv.xcomp_b);
        pB_x[m] = wm_ld(xhat, ownB * wv.xnode_b + wv.b_x + m * wv.xcomp_b);
      }
    }
    // ---- the first GR gradient rows (and the scalar gradient rows) are issued here, under the MFMAs
    float gx[GR][NM], gsv[HAS_S ? 16 : 1];
    auto load_group = [&](int r0) {
#pragma unroll
      for (int v0 = 0; v0 < GR; v0 += 4) {
        int g0[4];
        wm_tread<4>(trow, T_G0, r0 + v0, g0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t og = (uint32_t)g0[r] + wv.b_xe;
#pragma unroll
          for (int m = 0; m < NM; ++m) gx[v0 + r][m] = wm_ld(grad_x, og + 4u * m);
        }
      }
    };
    load_group(0);
    if constexpr (HAS_S) {
#pragma unroll
      for (int v0 = 0; v0 < 16; v0 += 4) {
        int g1[4];
        wm_tread<4>(trow, T_G1, v0, g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) gsv[v0 + r] = wm_ld(grad_s, (uint32_t)g1[r] + wv.b_s);
      }
    }
    float R[KS], Rd[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      R[s] = row.R[0][s];
      Rd[s] = row.R[1][s];
    }
    const WmIdx ixn = ix;
    const bool nfirst = row.first, nlast = row.last;
    (void)nfirst;
    (void)nlast;
    ix = wm_idx(a, lane, t + 2, e0, e1, e2);
    XEQ_WM_SB();
    float pd[16];
    WM_STAMP(0);   // tile top: table reads, gathers and prefetches issued
    {  // ---- pass S
      const f32x16 ds = wm_filter<KS>(R, Ws), qs = wm_filter<KS>(Rd, Ws);
      XEQ_WM_SB();
      WM_STAMP(1);   // MFMAs of pass S issued
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += GR) {
        if (r0 > 0) {
          XEQ_WM_SB();
          load_group(r0);
        }
#pragma unroll
        for (int c0 = r0; c0 < r0 + GR; c0 += 4) {
          const uint32_t any_first = (mk.first >> wm_bit(c0)) & 0xFFu, any_last = (mk.last >> wm_bit(c0)) & 0xFFu;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = c0 + r, gv = v - r0;
            if (any_first) {   // some half starts a segment in this chunk: load the new owner's rows
              if ((mfirst >> wm_bit(v)) & 1u) {
                if (v == vA) {
                  o_hs = pA_hs;
#pragma unroll
                  for (int m = 0; m < NM; ++m) o_x[m] = pA_x[m];
                } else if (v == vB) {
                  o_hs = pB_hs;
#pragma unroll
                  for (int m = 0; m < NM; ++m) o_x[m] = pB_x[m];
                } else {
                  const uint32_t own = (uint32_t)trow[T_OWN * 32 + v];
                  o_hs = wm_ld(h, own * row_h + wv.b_hs);
                  const uint32_t ox = own * wv.xnode_b + wv.b_x;
#pragma unroll
                  for (int m = 0; m < NM; ++m) o_x[m] = wm_ld(xhat, ox + m * wv.xcomp_b);
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) a_x[m] = 0.f;
                a_hs = 0.f;
              }
            }
            float dgs = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m) dgs += o_x[m] * gx[gv][m];
            const float ps = ds[v];
            a_hs += ps * dgs;
            const float gate = o_hs * ps;
#pragma unroll
            for (int m = 0; m < NM; ++m) a_x[m] += gate * gx[gv][m];
            pd[v] = o_hs * dgs * qs[v];
            if (any_last) {
              if ((mlast >> wm_bit(v)) & 1u) {
                const uint32_t own = (uint32_t)trow[T_OWN * 32 + v];
                wm_st(grad_h, own * row_h + wv.b_hs, a_hs);
                const uint32_t ox = own * wv.xnode_b + wv.b_x;
#pragma unroll
                for (int m = 0; m < NM; ++m) wm_st(grad_xhat, ox + m * wv.xcomp_b, a_x[m]);
              }
            }
          }
        }
      }
    }
    WM_STAMP(2);     // rows of pass S
    // the first rows of pass E fly under its MFMAs
    if (GR < 16) load_group(0);
    XEQ_WM_SB();
    const int my_slot = half_beg + 16 * t + keep_v;
    const bool keeper = keep_v >= 0 && my_slot < half_end;      // lanes 16..31 of the half whose row exists
    {  // ---- pass E
      const f32x16 de = wm_filter<KS>(R, We), qe = wm_filter<KS>(Rd, We);
      if constexpr (!HAS_S) {   // last MFMAs of the tile are issued: the next tile's records take over their registers
        XEQ_WM_SB();
        wm_row<KS, 2, (NM > 1)>(a, ixn, hh, rec, drec, row);
      }
      XEQ_WM_SB();
      WM_STAMP(3);   // MFMAs of pass E issued
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += GR) {
        if (r0 > 0) {
          XEQ_WM_SB();
          load_group(r0);
        }
#pragma unroll
        for (int c0 = r0; c0 < r0 + GR; c0 += 4) {
          float Yc[NM][4];
          if constexpr (NM > 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) wm_tread<4>(trow, T_Y + YOFF + m, c0, Yc[m]);
          }
          const uint32_t any_first = (mk.first >> wm_bit(c0)) & 0xFFu, any_last = (mk.last >> wm_bit(c0)) & 0xFFu;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = c0 + r, gv = v - r0;
            if (any_first) {
              if ((mfirst >> wm_bit(v)) & 1u) {
                o_he = v == vA ? pA_he : (v == vB ? pB_he : wm_ld(h, (uint32_t)trow[T_OWN * 32 + v] * row_h + wv.b_hs + he_off));
                a_he = 0.f;
              }
            }
            float dge = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m) dge += (NM > 1 ? Yc[m][r] : 1.f) * gx[gv][m];
            const float pe = de[v];
            a_he += pe * dge;
            pd[v] += o_he * dge * qe[v];
            if constexpr (NM > 1) {   // dL/dY_lm of the row's edge: sum over the unit's 32 channels
              const float wy = o_he * pe;
#pragma unroll
              for (int m = 0; m < NM; ++m) {
                const float ry = wm_half_total(wy * gx[gv][m]);
                if (writer) ysc[m * 32 + 16 * hh + v] = ry;     // parked in the wave's LDS line, stored after the pass
              }
            }
            if (any_last) {
              if ((mlast >> wm_bit(v)) & 1u) wm_st(grad_h, (uint32_t)trow[T_OWN * 32 + v] * row_h + wv.b_hs + he_off, a_he);
            }
          }
        }
      }
    }
    if constexpr (NM > 1) {
      // lane 31 of each half wrote ysc, lanes 16..31 read it: order the LDS accesses inside the wave
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (keeper) {
        float* dst = NM == 3 ? parts.y1 : parts.y2;
#pragma unroll
        for (int m = 0; m < NM; ++m) dst[((int64_t)un.cb * NM + m) * E + my_slot] = ysc[m * 32 + 16 * hh + keep_v];
      }
    }
    WM_STAMP(4);     // rows of pass E
    if constexpr (HAS_S) {  // ---- pass M
      const f32x16 dm = wm_filter<KS>(R, Wm), qm = wm_filter<KS>(Rd, Wm);
      XEQ_WM_SB();
      wm_row<KS, 2, (NM > 1)>(a, ixn, hh, rec, drec, row);
      WM_STAMP(5);   // MFMAs of pass M issued
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += 4) {
        const uint32_t any_first = (mk.first >> wm_bit(c0)) & 0xFFu, any_last = (mk.last >> wm_bit(c0)) & 0xFFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = c0 + r;
          if (any_first) {
            if ((mfirst >> wm_bit(v)) & 1u) {
              o_hm = v == vA ? pA_hm : (v == vB ? pB_hm : wm_ld(h, (uint32_t)trow[T_OWN * 32 + v] * row_h + wv.b_hm));
              a_hm = 0.f;
            }
          }
          a_hm += dm[v] * gsv[v];
          pd[v] += o_hm * gsv[v] * qm[v];
          if (any_last) {
            if ((mlast >> wm_bit(v)) & 1u) wm_st(grad_h, (uint32_t)trow[T_OWN * 32 + v] * row_h + wv.b_hm, a_hm);
          }
        }
      }
    }
    WM_STAMP(6);     // rows of pass M
    // ---- dL/dd of every row's edge: sum over the unit's 32 channels
    float kp = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float tot = wm_half_total(pd[v]);
      kp = keep_v == v ? tot : kp;
    }
    if (keeper) parts.pd[(int64_t)unit * E + my_slot] = kp;
    mk = wm_table<KS, 2, (NM > 1)>(lane, ixn, row, stride0, stride1, tnext);
    __builtin_amdgcn_wave_barrier();
    WM_STAMP(7);     // channel sums, partial writes, next tile's table
  }
#ifdef XEQ_WM_STAMPS
  if (HAS_S && lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_wm_stamps[i], st_[i]);
#endif
}

template <int KS>
__global__ void __launch_bounds__(64 * WM_WAVES) __attribute__((amdgpu_waves_per_eu(XEQ_WM_BWD_WPE))) k_message_bwd_wm(WmArgs a, const float* __restrict__ rec,
                                                                  const float* __restrict__ drec, const float* __restrict__ h,
                                                                  const float* __restrict__ xhat, const float* __restrict__ grad_s,
                                                                  const float* __restrict__ grad_x, const float* __restrict__ w_rbf,
                                                                  const float* __restrict__ b_rbf, float* __restrict__ grad_h,
                                                                  float* __restrict__ grad_xhat, WmParts parts) {
  __shared__ int tbl_all[WM_WAVES][2 * T_ROWS * 32];
  __shared__ float wl[3 * KS * 64];
  __shared__ float ysc_all[WM_WAVES][5 * 32];            // per wave: dL/dY_lm totals of a tile's rows (l > 0 units)
  int range, unit;
  wm_decode(a, a.nu[0] + a.nu[1] + a.nu[2], range, unit);
  const WmUnit un = wm_unit(a, unit);
  wm_stage_weights<KS>(a, un, w_rbf, b_rbf, wl);
  __syncthreads();
  if (range >= a.n_ranges) return;
  int* tbl = tbl_all[threadIdx.x >> 6];
  float* ysc = ysc_all[threadIdx.x >> 6];
#ifdef XEQ_WM_ONLY_L
  if (un.l == XEQ_WM_ONLY_L)
    wm_bwd_body<2 * XEQ_WM_ONLY_L + 1, KS>(a, range, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, ysc);
#else
  if (un.l == 0) wm_bwd_body<1, KS>(a, range, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, ysc);
  else if (un.l == 1) wm_bwd_body<3, KS>(a, range, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, ysc);
  else wm_bwd_body<5, KS>(a, range, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, ysc);
#endif
}

// stream_ptr[k] = first node c with rowptr[c] >= k E / (2 R)  (k < 2R), stream_ptr[2R] = N: streams of about equal
// edge counts whose boundaries sit on segment starts; every node belongs to exactly one stream
__global__ void k_wm_stream_ptr(const int32_t* __restrict__ rowptr, int64_t N, int64_t E, int n_ranges, int32_t* __restrict__ sp) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > 2 * n_ranges) return;
  if (k == 2 * n_ranges) {
    sp[k] = (int32_t)N;
    return;
  }
  const int64_t target = (int64_t)k * E / (2 * n_ranges);
  int64_t lo = 0, hi = N;   // first c in [0, N] with rowptr[c] >= target (rowptr[N] = E >= target)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] >= target) hi = mid;
    else lo = mid + 1;
  }
  sp[k] = (int32_t)lo;
}

// dL/dvec from the per-unit partials (indexed by slot of the reverse walk order), summed in unit order (deterministic),
// chain rule of A1-A3
__global__ void k_wm_edge_grad(const float* __restrict__ vec, int64_t E, int nu, int nu1, int nu2, const float* __restrict__ pd,
                               const float* __restrict__ y1, const float* __restrict__ y2, const int32_t* __restrict__ perm,
                               float* __restrict__ grad_vec) {
  const int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // slot of the reverse walk order
  if (sl >= E) return;
  const int64_t e = perm ? perm[sl] : sl;
  float gd = 0.f, q1[3] = {0.f, 0.f, 0.f}, q2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int u = 0; u < nu; ++u) gd += pd[(int64_t)u * E + sl];
  for (int u = 0; u < nu1; ++u)
#pragma unroll
    for (int m = 0; m < 3; ++m) q1[m] += y1[((int64_t)u * 3 + m) * E + sl];
  for (int u = 0; u < nu2; ++u)
#pragma unroll
    for (int m = 0; m < 5; ++m) q2[m] += y2[((int64_t)u * 5 + m) * E + sl];
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

// sizes the 32-bit byte offsets of the kernels cover (rows of h: n_nodes * H * 4 bytes; records: n_edges * 288 bytes)
static bool wm_fits(int64_t n_nodes, int64_t n_edges, int node_dim, const int32_t mul[3]) {
  const int64_t H = node_dim + 2 * (int64_t)(mul[0] + mul[1] + mul[2]);
  return n_nodes >= 0 && n_edges >= 0 && n_nodes * H * 4 < (1ll << 32) && n_edges * (int64_t)72 * 4 < (1ll << 32);
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
  XEQ_CHECK_ARG(wm_fits(n_nodes, n_edges, node_dim, mul), "%s: tensors too large for 32-bit byte offsets (shard the batch)", who);
  for (int l = 0; l < 3; ++l) a.nu[l] = mul[l] / 32;
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.n_ranges = n_ranges;
  return XEQ_OK;
}

}  // namespace xeq

using namespace xeq;

// KS covers K = B + 1 (bias column) in steps of two; records are written for the same bucket
#define XEQ_WM_DISPATCH(KERNEL, ...)                                                                                   \
  do {                                                                                                                 \
    const int ks = wm_ks(num_basis);                                                                                   \
    if (ks <= 5) hipLaunchKernelGGL((KERNEL<5>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__);      \
    else if (ks <= 9) hipLaunchKernelGGL((KERNEL<9>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (ks <= 11) hipLaunchKernelGGL((KERNEL<11>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<16>), grid, dim3(64 * WM_WAVES), 0, (hipStream_t)stream, __VA_ARGS__);             \
  } while (0)

// the template KS a given num_basis is dispatched to; the records are laid out for it
static int wm_ks_bucket(int num_basis) {
  const int ks = wm_ks(num_basis);
  return ks <= 5 ? 5 : (ks <= 9 ? 9 : (ks <= 11 ? 11 : 16));
}
static int wm_kp_bucket(int num_basis) { return (wm_ks_bucket(num_basis) + 3) & ~3; }
static unsigned wm_grid(int n_ranges, int nunits) {
  int64_t blocks = (int64_t)((n_ranges + WM_WAVES - 1) / WM_WAVES) * nunits;   // one workgroup per (range group, unit)
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

int xeq_message_wm_fits(int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]) {
  return wm_supported(num_basis, node_dim, mul) && wm_fits(n_nodes, n_edges, node_dim, mul) ? 1 : 0;
}

int xeq_message_wm_streams(const int32_t* rowptr, int64_t n_nodes, int64_t n_edges, int n_ranges, int32_t* stream_ptr,
                           void* stream) {
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_ranges >= 1, "xeq_message_wm_streams: bad sizes");
  const int n = 2 * n_ranges + 1;
  hipLaunchKernelGGL(k_wm_stream_ptr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rowptr, n_nodes,
                     n_edges, n_ranges, stream_ptr);
  XEQ_CHECK_LAUNCH("xeq_message_wm_streams");
  return XEQ_OK;
}

int xeq_message_fwd_wm(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* stream_ptr, const int32_t* c_rowptr,
                       const int32_t* c_perm, const int64_t* center, const int64_t* nbr, const void* basis,
                       const void* h, const void* xhat, const void* s_in, const void* x_in, const void* w_rbf,
                       const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* s_out, void* x_out,
                       int xhat_layout, void* stream) {
  WmArgs a{};
  int rcode = wm_check("xeq_message_fwd_wm", n_nodes, n_edges, n_ranges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.stream_ptr = stream_ptr;
  a.rowptr = c_rowptr;
  a.perm = c_perm;
  a.owner = center;
  a.gather = nbr;
  a.xl = xhat_layout & 1;   // the XEQ_XHAT_HIGHER_L_ZERO hint is for the wq kernels; this family computes the general form
  XEQ_CHECK_ARG(n_ranges >= 1, "%s: the stream table must cover every node (n_ranges >= 1)", "xeq_message_wm");
  const int nunits = a.nu[0] + a.nu[1] + a.nu[2];
  dim3 grid(wm_grid(n_ranges, nunits));
  XEQ_WM_DISPATCH(k_message_fwd_wm, a, (const float*)basis, (const float*)h, (const float*)xhat, (const float*)s_in,
                  (const float*)x_in, (const float*)w_rbf, (const float*)b_rbf, (float*)s_out, (float*)x_out);
  XEQ_CHECK_LAUNCH("xeq_message_fwd_wm");
  return XEQ_OK;
}

int xeq_message_bwd_wm(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* stream_ptr, const int32_t* n_rowptr,
                       const int32_t* n_perm, const int64_t* center, const int64_t* nbr, const void* basis,
                       const void* dbasis, const void* h, const void* xhat, const void* grad_s, const void* grad_x,
                       const void* w_rbf, const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* grad_h,
                       void* grad_xhat, void* parts, int xhat_layout, void* stream) {
  WmArgs a{};
  int rcode = wm_check("xeq_message_bwd_wm", n_nodes, n_edges, n_ranges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.stream_ptr = stream_ptr;
  a.rowptr = n_rowptr;
  a.perm = n_perm;
  a.owner = nbr;
  a.gather = center;
  a.xl = xhat_layout & 1;   // the XEQ_XHAT_HIGHER_L_ZERO hint is for the wq kernels; this family computes the general form
  XEQ_CHECK_ARG(n_ranges >= 1, "%s: the stream table must cover every node (n_ranges >= 1)", "xeq_message_wm");
  const int nunits = a.nu[0] + a.nu[1] + a.nu[2];
  WmParts pr;
  pr.pd = (float*)parts;
  pr.y1 = pr.pd + (int64_t)nunits * n_edges;
  pr.y2 = pr.y1 + (int64_t)a.nu[1] * 3 * n_edges;
  dim3 grid(wm_grid(n_ranges, nunits));
  XEQ_WM_DISPATCH(k_message_bwd_wm, a, (const float*)basis, (const float*)dbasis, (const float*)h, (const float*)xhat,
                  (const float*)grad_s, (const float*)grad_x, (const float*)w_rbf, (const float*)b_rbf, (float*)grad_h,
                  (float*)grad_xhat, pr);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_wm");
  return XEQ_OK;
}

/* development: read and clear the phase cycle counters of a -DXEQ_WM_STAMPS build */
int xeq_wm_debug_stamps(unsigned long long out[8]) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wm_stamps), sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wm_stamps), zero, sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  return XEQ_OK;
}

/* floats of the `parts` scratch buffer of xeq_message_bwd_wm */
int64_t xeq_message_wm_parts_floats(int64_t n_edges, const int32_t mul[3]) {
  return n_edges * (int64_t)(mul[0] / 32 + mul[1] / 32 + mul[2] / 32 + 3 * (mul[1] / 32) + 5 * (mul[2] / 32));
}

int xeq_message_wm_edge_grad(const void* vec, int64_t n_edges, const int32_t mul[3], const void* parts, const int32_t* n_perm,
                             void* grad_vec, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && mul[0] % 32 == 0 && mul[1] % 32 == 0 && mul[2] % 32 == 0, "xeq_message_wm_edge_grad: bad sizes");
  if (n_edges == 0) return XEQ_OK;
  const int nu1 = mul[1] / 32, nu2 = mul[2] / 32, nunits = mul[0] / 32 + nu1 + nu2;
  const float* pd = (const float*)parts;
  const float* y1 = pd + (int64_t)nunits * n_edges;
  const float* y2 = y1 + (int64_t)nu1 * 3 * n_edges;
  hipLaunchKernelGGL(k_wm_edge_grad, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)vec, n_edges, nunits, nu1, nu2, pd, y1, y2, n_perm, (float*)grad_vec);
  XEQ_CHECK_LAUNCH("xeq_message_wm_edge_grad");
  return XEQ_OK;
}

}  // extern "C"
