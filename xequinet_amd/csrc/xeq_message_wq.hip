// Fused XPaiNN message kernels, "wave / quad" form (fp32; the default where the channel layout allows it).
// Reference dataflow: nn/xpainn.py:140-159; reverse pass for nn/basic.py:143-159.
//
// Arithmetic mapping (kept from round 1's wave / matrix-core form, xeq_message_wm.hip, retired in round 4): the filter
// phi_e = (W rho(d_e) + b) f(d_e) is a matrix-core tile D[edge][channel] (K = B + 1),
// a wave owns (range of nodes, 32 gate channels of one l), its two half-waves are two independent streams over
// contiguous CSR segments, and the sum over the edges of a node is a running sum in registers that is stored once.
//
// The WALK ORDER removes the per-row bookkeeping that bounded that form (19 vector
// instructions per MFMA, 244 VGPRs, a divergent first/last branch per row):
//
//   * every node's edge list is padded to a multiple of FOUR slots ("quads"; padding slots carry an all-zero record,
//     so their filter is exactly 0).  The four accumulator registers 4g .. 4g+3 of a lane are then the four rows of
//     ONE quad, and a quad belongs to ONE node: segment starts and ends can only fall between quads.  The rows of a
//     quad are plain FMAs into a quad sum; the segment logic (reset / residual / the node's only store) runs once
//     per quad on half-uniform flags, 4x less often and branch-free except for the store.
//   * the walk plan (xeq_message_wq_plan) is built once per graph, independent of the positions: per padded slot the
//     gathered node and the edge id, per quad the owner node and its first / last flags, stream boundaries on quads.
//     The per-edge records (xeq_edge_basis_wq) are written IN PADDED WALK ORDER, so a tile's A operand, its Y_lm and
//     its indices are sequential loads with no dependent index chain.
//   * what a whole quad shares comes from its owner: in the reverse pass the owner's rows (h, xhat) are per-quad
//     constants, so products with them are hoisted out of the rows (l = 0: 7 vector instructions per row).
//   * the per-edge sums over channels of the reverse pass (dL/dd, dL/dY_lm) use a register-halving DPP butterfly:
//     16 rows x 32 lanes reduce in 48 instead of 80 cross-lane adds, and leave as one 64-byte store per quantity.
//
// This source is compiled TWICE (csrc/build.py): as it stands for the walk plan, the records and the forward kernel (LLVM's
// max-ILP machine scheduler: forward launch 201 against 212 us), and through xeq_message_wq_bwd.hip with XEQ_WQ_PART_BWD
// defined for the reverse kernel and the edge gradients (default scheduler, which weighs register pressure: reverse launch
// 355 against 368 us).  Everything above the kernels is shared text.
#include "xeq_common.h"

#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace xeq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// The filter contraction, split (round 3).  phi[e][ch] = sum_k rho~[e][k] W[ch][k] was K + 1 = 21 exact-f32 MFMA steps (11 x
// v_mfma_f32_32x32x2_f32 = 704 matrix-pipe cycles per 32 x 32 tile), and the tile loops of these kernels run within 80 % of that
// pipe bound (two waves share a SIMD's pipe; step timeline of round 3).  The f32 MFMA runs at 1/16 of the bf16 rate, so the first
// sixteen k now go through bf16 MFMAs on operands split three ways, x = hi + mid + lo with bf16 parts (8 + 8 + 8 significant
// bits, exact): six products hi hi, hi mid, mid hi, hi lo, lo hi, mid mid of v_mfma_f32_32x32x16_bf16 (32 cycles each, f32
// accumulate) leave out mid lo, lo mid, lo lo = 2^-24 of a product, the size of one f32 rounding; the remaining k (16 .. B - 1)
// and the bias column stay exact-f32 steps.  K = 21: 6 x 32 + 3 x 64 = 384 pipe cycles instead of 704.
// dwords per record: [kh][hi | mid | lo][4] bf16 packs of k = 8 kh .. 8 kh + 7 (24), [kh][TW] f32 tail (q = 2 s + kh -> k = 16 + q, then the
// envelope for the bias; TW = 4 tail values per k half up to 23 basis functions (KS <= 4 exact-f32 steps), 8 up to 31 (KS = 8)), Y1[3] Y2[5]:
// 40 dwords (160 B) or 48 (192 B)
// Round 6: the tail of up to FIVE values (k = 16 .. 19 and the bias column: num_basis <= 20, the default) runs through the bf16 matrix
// instruction as well.  A 16-slot k block holds the three splits of the five values, A = [hi_0..4 | mid_0..4 | lo_0..4 | 0], and is
// multiplied against THREE weight arrangements B1 = [hi | hi | hi | 0], B2 = [mid | mid | 0 | 0], B3 = [lo | 0 | 0 | 0] (packed once per
// weight version): A B1 + A B2 + A B3 = a_hi (b_hi + b_mid + b_lo) + a_mid (b_hi + b_mid) + a_lo b_hi, the same six products as the
// first sixteen k (what is left out is 2^-24 of a product).  Three 32-cycle instructions replace three 64-cycle exact-f32 steps: 288
// instead of 384 matrix-pipe cycles per filter tile; the record keeps its 160 bytes (the tail's eight floats become eight dwords of
// bf16 pairs).  Wider tails (num_basis 21 .. 31) keep the exact-f32 steps.
constexpr bool wq_bftail(int ks) { return ks <= 3; }
constexpr int WQ_TAIL = 24;                // dword offset of the tail inside a record
constexpr int wq_tailw(int ks) { return ks <= 4 ? 4 : 8; }
constexpr int wq_recf(int ks) { return WQ_TAIL + 2 * wq_tailw(ks) + 8; }
constexpr int wq_yoff(int ks) { return WQ_TAIL + 2 * wq_tailw(ks); }
// KS: 1 / 3 = a tail of up to 1 / 5 values (basis functions from k = 16 on, then the bias column) in the bf16 block (wq_bftail); from six
// values on, the exact-f32 steps of the tail, two values per step (never below 4, so that the two forms do not share a KS)
constexpr int wq_tail_values(int num_basis) { return (num_basis > 16 ? num_basis - 16 : 0) + 1; }
constexpr int wq_ks(int num_basis) {
  return wq_tail_values(num_basis) <= 1 ? 1 : (wq_tail_values(num_basis) <= 5 ? 3 : ((wq_tail_values(num_basis) + 1) / 2 < 4 ? 4 : (wq_tail_values(num_basis) + 1) / 2));
}
constexpr int WQ_REC_MAX = 48;
__device__ __forceinline__ uint32_t wq_bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x = hi + mid + lo, each a bf16 (round to nearest even of what is left): returns the three 16-bit patterns
__device__ __forceinline__ void wq_split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = wq_bf16_rne(x);
  const float r1 = x - __uint_as_float(hi << 16);
  mid = wq_bf16_rne(r1);
  const float r2 = r1 - __uint_as_float(mid << 16);
  lo = wq_bf16_rne(r2);
}
// value at tail position q of a record / weight row: k = 16 + q while k < B, then the bias column, then zeros
__device__ __forceinline__ int wq_tail_k(int q, int B) { return 16 + q < B ? 16 + q : (16 + q == (B > 16 ? B : 16) ? -1 : -2); }
constexpr uint32_t WQ_FIRST = 1u << 30, WQ_LAST = 1u << 31, WQ_OWNER = (1u << 30) - 1u;
#ifndef XEQ_WQ_WAVES
#define XEQ_WQ_WAVES 4
#endif
constexpr int WQ_WAVES = XEQ_WQ_WAVES;     // waves per workgroup: they share the unit's weights and, step by step, the window

#ifndef XEQ_WQ_PART_BWD   // the forward half of this file (see the note at its top): plan, records, forward kernel
// ------------------------------------------------------------------------------------------------ walk plan
struct QuadCount {
  const int32_t* rowptr;
  int64_t n;
  // a node without an edge still gets ONE quad (four padding slots: zero records, gathering its own rows): it then takes the
  // ordinary path -- its running sums are exact zeros, its store writes s_out = s_in / zero gradients -- and the plan spreads such
  // nodes over the ranges by their quads.  (They used to be picked up range by range in a serial loop of one wave: the 400 padding
  // atoms of a capacity-sized batch sat behind the LAST range and cost 5 % of a step, 2 000 of them doubled it.)
  __host__ __device__ int32_t operator()(int64_t i) const {
    if (i >= n) return 0;
    const int32_t deg = rowptr[i + 1] - rowptr[i];
    return deg > 0 ? (deg + 3) >> 2 : 1;
  }
};
#ifndef XEQ_WQ_PART_BWD
// qptr[0 .. n] = exclusive prefix sums of the quads per node, by one workgroup (xeq_common.h: wg_scan_lds)
__global__ void __launch_bounds__(SCAN_WG_THREADS) k_wq_quad_scan(const QuadCount op, int32_t* __restrict__ qptr) {
  extern __shared__ int32_t scan_lds[];
  const int32_t total = wg_scan_lds(op, op.n, scan_lds);
#pragma unroll 4
  for (int64_t i = threadIdx.x; i < op.n; i += SCAN_WG_THREADS) qptr[i] = scan_lds[i];
  if (threadIdx.x == 0) qptr[op.n] = total;
}
#endif

// one thread per slot of the walk order: its padded position, the pads behind a segment's last slot, the quad records
__global__ void k_wq_fill(int64_t E, int64_t n_nodes, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                          const int64_t* __restrict__ owner, const int64_t* __restrict__ gather,
                          const int32_t* __restrict__ qptr, int32_t* __restrict__ pgath, int32_t* __restrict__ peid,
                          uint32_t* __restrict__ qinfo) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= E) {   // threads E .. E + N - 1: the lone quad of a node without an edge
    const int64_t n = s - E;
    if (n < n_nodes && rowptr[n] == rowptr[n + 1]) {
      const int64_t q0 = qptr[n];
      for (int i = 0; i < 4; ++i) {
        pgath[4 * q0 + i] = (int32_t)n;
        peid[4 * q0 + i] = -1;
      }
      qinfo[q0] = (uint32_t)n | WQ_FIRST | WQ_LAST;
    }
    return;
  }
  if (s >= rowptr[n_nodes]) return;   // E may be a capacity: the walk ends at rowptr[N] (device-side edge count)
  const int32_t eid = perm ? perm[s] : (int32_t)s;
  const int64_t n = owner[eid];
  const int32_t r0 = rowptr[n], deg = rowptr[n + 1] - r0, k = (int32_t)(s - r0), q0 = qptr[n], nq = (deg + 3) >> 2;
  const int64_t p = 4 * (int64_t)q0 + k;
  pgath[p] = (int32_t)gather[eid];
  peid[p] = eid;
  if (k == deg - 1)
    for (int64_t pp = p + 1; pp < 4 * (int64_t)(q0 + nq); ++pp) {   // pads: zero record, gather the owner's own rows
      pgath[pp] = (int32_t)n;
      peid[pp] = -1;
    }
  if ((k & 3) == 0) {
    const int32_t g = k >> 2;
    qinfo[q0 + g] = (uint32_t)n | (g == 0 ? WQ_FIRST : 0u) | (g == nq - 1 ? WQ_LAST : 0u);
  }
}

// sq[k] / sn[k]: first quad / node of stream k; boundaries sit on segment starts at about equal quad counts
__global__ void k_wq_streams(const int32_t* __restrict__ qptr, int64_t N, int n_ranges, int32_t* __restrict__ sq,
                             int32_t* __restrict__ sn) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > 2 * n_ranges) return;
  const int64_t Q = qptr[N];
  if (k == 2 * n_ranges) {
    sq[k] = (int32_t)Q;
    sn[k] = (int32_t)N;
    return;
  }
  const int64_t target = (int64_t)k * Q / (2 * n_ranges);
  int64_t lo = 0, hi = N;   // first n in [0, N] with qptr[n] >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (qptr[mid] >= target) hi = mid;
    else lo = mid + 1;
  }
  sq[k] = qptr[lo];
  sn[k] = (int32_t)lo;
}

// win[2 s], win[2 s + 1]: first gathered node and number of rows of the WINDOW of step s = the WQ_WAVES consecutive
// ranges one workgroup walks together: every node its 2 WQ_WAVES streams gather from lies in [w0, w0 + rows).  One wave
// per step.  For batches of molecules the window is the few molecules the step touches.
// mult: the stream CLASS (wq_long_mult below): a stream of the class is `mult` consecutive streams of the table
__global__ void k_wq_windows(const int32_t* __restrict__ sq, const int32_t* __restrict__ pgath, int n_ranges, int n_steps, int mult,
                             int32_t* __restrict__ win) {
  const int step = blockIdx.x, lane = threadIdx.x;
  if (step >= n_steps) return;
  const int k0 = min(2 * WQ_WAVES * mult * step, 2 * n_ranges), k1 = min(k0 + 2 * WQ_WAVES * mult, 2 * n_ranges);
  const int64_t p0 = 4 * (int64_t)sq[k0], p1 = 4 * (int64_t)sq[k1];
  int lo = 0x7fffffff, hi = -1;
  for (int64_t p = p0 + lane; p < p1; p += 64) {
    const int g = pgath[p];
    lo = g < lo ? g : lo;
    hi = g > hi ? g : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if (lane == 0) {
    win[2 * step] = hi >= 0 ? lo : 0;
    win[2 * step + 1] = hi >= 0 ? hi - lo + 1 : 0;
  }
}

// records in padded walk order (layout: WQ_REC above); val(k) = f rho_k, the bias column's multiplier is f; derivative record
// likewise.  Six threads per slot: two (one per k half) evaluate eight basis functions each and store their three bf16 packs,
// two the f32 tail, two the harmonics.
// one piece (grp: 0-1 bf16 packs of k half grp; 2-3 f32 tail of k half grp - 2; 4-5 harmonics) of the record of padded slot p, written
// through out / dout (the slot's record in the workgroup's LDS staging rows)
__device__ __forceinline__ void wq_record_piece(const float* __restrict__ vec, const int32_t* __restrict__ peid, int64_t p, int grp,
                                                const RadialSpec& rs, const float* __restrict__ p0, const float* __restrict__ p1,
                                                float* __restrict__ out, float* __restrict__ dout, int tailw, bool bftail) {
  const int yoff = WQ_TAIL + 2 * tailw;
  const int32_t e = peid[p];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (e < 0) {   // padding slot: an all-zero record (its filter is exactly 0)
    if (grp < 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        *reinterpret_cast<f32x4*>(out + 12 * grp + 4 * c) = zero;
        if (dout) *reinterpret_cast<f32x4*>(dout + 12 * grp + 4 * c) = zero;
      }
    } else if (grp < 4) {
      for (int c = 0; c < tailw; c += 4) {
        *reinterpret_cast<f32x4*>(out + WQ_TAIL + tailw * (grp - 2) + c) = zero;
        if (dout) *reinterpret_cast<f32x4*>(dout + WQ_TAIL + tailw * (grp - 2) + c) = zero;
      }
    } else {
      *reinterpret_cast<f32x4*>(out + yoff + 4 * (grp - 4)) = zero;
      if (dout) *reinterpret_cast<f32x4*>(dout + yoff + 4 * (grp - 4)) = zero;
    }
    return;
  }
  const float rc = (float)rs.cutoff;
  const EdgeGeom<float> g = edge_geom<float>(vec[3 * (int64_t)e], vec[3 * (int64_t)e + 1], vec[3 * (int64_t)e + 2]);
  const int B = rs.num_basis;
  if (grp >= 4) {
    float y1[3], y2[5];
    sph_harm_l12<float>(g, y1, y2);
    const f32x4 v = grp == 4 ? f32x4{y1[0], y1[1], y1[2], y2[0]} : f32x4{y2[1], y2[2], y2[3], y2[4]};
    *reinterpret_cast<f32x4*>(out + yoff + 4 * (grp - 4)) = v;
    if (dout) *reinterpret_cast<f32x4*>(dout + yoff + 4 * (grp - 4)) = v;   // the reverse kernel reads Y next to the derivatives (one record
    return;                                                                  // stream less per row load; measured: no change in traffic or
                                                                             // time -- the l > 0 units need the value record anyway, for the
                                                                             // gate_edge filter in dL/dY)
  }
  float f, df;
  envelope<float>(rs.cutoff_kind, g.d, rc, f, df);
  auto value = [&](int k, float& val, float& dval) {   // k >= 0: basis function k; -1: the bias column; -2: nothing
    val = dval = 0.f;
    if (k >= 0 && k < B) {
      float rho, drho;
      radial<float>(rs.rbf_kind, g.d, rc, p0[k], p1 ? p1[k] : 0.f, rho, drho, k, B);
      val = f * rho;
      dval = df * rho + f * drho;
    } else if (k == -1) {
      val = f;
      dval = df;
    }
  };
  if (grp < 2) {
    const int kh = grp;
    uint32_t w[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, dw[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      float val, dval;
      value(8 * kh + jj < B ? 8 * kh + jj : -2, val, dval);
      uint32_t a3[3], d3[3];
      wq_split3(val, a3[0], a3[1], a3[2]);
      wq_split3(dval, d3[0], d3[1], d3[2]);
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) {
        w[sp][jj >> 1] |= a3[sp] << (16 * (jj & 1));
        dw[sp][jj >> 1] |= d3[sp] << (16 * (jj & 1));
      }
    }
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      *reinterpret_cast<f32x4*>(out + 12 * kh + 4 * sp) =
          f32x4{__uint_as_float(w[sp][0]), __uint_as_float(w[sp][1]), __uint_as_float(w[sp][2]), __uint_as_float(w[sp][3])};
      if (dout)
        *reinterpret_cast<f32x4*>(dout + 12 * kh + 4 * sp) =
            f32x4{__uint_as_float(dw[sp][0]), __uint_as_float(dw[sp][1]), __uint_as_float(dw[sp][2]), __uint_as_float(dw[sp][3])};
    }
  } else if (bftail) {   // slots 8 kh .. 8 kh + 7 of [hi_0..4 | mid_0..4 | lo_0..4 | 0]: the three bf16 parts of the five tail values
    const int kh = grp - 2;
    uint32_t a3[5][3], d3[5][3];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float val, dval;
      value(wq_tail_k(q, B), val, dval);
      wq_split3(val, a3[q][0], a3[q][1], a3[q][2]);
      wq_split3(dval, d3[q][0], d3[q][1], d3[q][2]);
    }
    uint32_t w[4] = {0u, 0u, 0u, 0u}, dw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int sl0 = jj, sl1 = 8 + jj;   // (both k halves unrolled, one selected: the slot arithmetic stays compile-time)
      const uint32_t v0 = sl0 < 15 ? a3[sl0 % 5][sl0 / 5] : 0u, v1 = sl1 < 15 ? a3[sl1 % 5][sl1 / 5] : 0u;
      const uint32_t e0 = sl0 < 15 ? d3[sl0 % 5][sl0 / 5] : 0u, e1 = sl1 < 15 ? d3[sl1 % 5][sl1 / 5] : 0u;
      w[jj >> 1] |= (kh ? v1 : v0) << (16 * (jj & 1));
      dw[jj >> 1] |= (kh ? e1 : e0) << (16 * (jj & 1));
    }
    *reinterpret_cast<f32x4*>(out + WQ_TAIL + 4 * kh) = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
    if (dout)
      *reinterpret_cast<f32x4*>(dout + WQ_TAIL + 4 * kh) = f32x4{__uint_as_float(dw[0]), __uint_as_float(dw[1]), __uint_as_float(dw[2]), __uint_as_float(dw[3])};
  } else {
    const int kh = grp - 2;
    for (int c0 = 0; c0 < tailw; c0 += 4) {   // tail value s of this k half: position q = 2 s + kh
      f32x4 v, dv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float val, dval;
        value(wq_tail_k(2 * (c0 + c) + kh, B), val, dval);
        v[c] = val;
        dv[c] = dval;
      }
      *reinterpret_cast<f32x4*>(out + WQ_TAIL + tailw * kh + c0) = v;
      if (dout) *reinterpret_cast<f32x4*>(dout + WQ_TAIL + tailw * kh + c0) = dv;
    }
  }
}


// A workgroup writes the records of WQ_REC_SLOTS consecutive padded slots: waves 0 / 1 / 2 compute the bf16 packs / the f32 tails / the
// harmonics of all of them (one KIND of piece per wave: with the six pieces of a slot on six neighbouring lanes every wave ran all three
// branches one after the other at a third of its lanes) into LDS rows, then all 256 threads store the rows with whole 16-byte lanes
// (the pieces are 16-48 bytes at a stride of 160: stored directly they kept the kernel at 2.6 TB/s of writes).  48.5 -> 42 us with the
// wave-uniform kinds alone; the same values bit for bit.
constexpr int WQ_REC_SLOTS = 32;
__global__ void __launch_bounds__(256) k_wq_records(const float* __restrict__ vec, const int32_t* __restrict__ peid,
                             const int32_t* __restrict__ qptr, int64_t N, int64_t pcap, RadialSpec rs,
                             const float* __restrict__ p0, const float* __restrict__ p1, float* __restrict__ rec,
                             float* __restrict__ drec, int recf, int tailw, int bftail) {
  __shared__ __attribute__((aligned(16))) float stage[2][WQ_REC_SLOTS * 48];
  const int64_t limit = min(pcap, 4 * (int64_t)qptr[N]);
  const int64_t first = (int64_t)blockIdx.x * WQ_REC_SLOTS;
  if (first >= limit) return;                                   // uniform
  const int t = threadIdx.x, kind = t >> 6, r = t & 63, slot = r >> 1;
  const int64_t p = first + slot;
  if (kind < 3 && p < limit)
    wq_record_piece(vec, peid, p, 2 * kind + (r & 1), rs, p0, p1, stage[0] + slot * recf, drec ? stage[1] + slot * recf : nullptr, tailw, bftail != 0);
  __syncthreads();
  const int n4 = (int)min((int64_t)WQ_REC_SLOTS, limit - first) * recf / 4;
  const f32x4* s0 = reinterpret_cast<const f32x4*>(stage[0]);
  const f32x4* s1 = reinterpret_cast<const f32x4*>(stage[1]);
  f32x4* g0 = reinterpret_cast<f32x4*>(rec + first * recf);
  f32x4* g1 = drec ? reinterpret_cast<f32x4*>(drec + first * recf) : nullptr;
  for (int i = t; i < n4; i += 256) {
    g0[i] = s0[i];
    if (g1) g1[i] = s1[i];
  }
}

#endif
// ------------------------------------------------------------------------------------------------ common
struct WqArgs {
  int64_t n_nodes, n_edges, pcap;
  int n_ranges;
  const int32_t* sq;       // [2 R + 1] quad boundaries of the streams (range w: streams 2w and 2w + 1)
  const int32_t* sn;       // [2 R + 1] node boundaries of the streams
  const int32_t* rowptr;   // [N + 1] CSR of the walk order (isolated nodes)
  const int32_t* pgath;    // [P] gathered node per padded slot
  const uint32_t* qinfo;   // [Q] owner | WQ_FIRST | WQ_LAST per quad
  const int32_t* win;      // [2 n_steps(1)] window (first node, rows) of every step of the SHORT class, then [2 n_steps(mlong)] of the long class
  // Stream classes (round 6).  The stream table (sq, sn) holds SHORT streams; the l = 0 units walk LONG streams, each `mlong` consecutive
  // table streams (1: one class, as up to round 5).  Why: the l > 0 units gather 5-7 pieces of 128 bytes per window row, so their steps
  // must stay short for the window to fit LDS (a step that does not fit gathers from global memory: the l = 2 unit's forward launch alone
  // 90 us at 48 edges per stream, 148 us at 80), while the l = 0 units -- 2-4 pieces per row, more than half of the work -- run fastest on
  // long streams (fewer range prologues, window stagings and barriers per edge).  The work of a workgroup is cut in UNITS of `mlong`
  // short steps = one long step, so that the units of one chunk still walk the same records at about the same time.
  int mlong;
  int n_steps, steps_per_wg;   // in chunk UNITS (see above); steps_per_wg: units of a LONG chunk
  int regions, region_steps;   // the units are cut in `regions` contiguous regions (one per XCD when the grid is XCD-mapped)
  int lvl_chunks[3], lvl_spw[3], lvl_start[3];   // per region: three runs of chunks, long to short (chunks, steps per chunk, first
                                                 // step): the grid ends on short workgroups (a region's last chunks start last)
  int F, C, D, H, B;
  Irreps ir;
  int xl;                  // layout of xhat / grad_xhat
  int mirror;              // reverse pass over the FORWARD plan of a symmetric list: every slot stands for its mirror edge (Y_1 negated)
  int packed_w;            // w_rbf points at xeq_message_wq_pack_weights' output (XEQ_WQ_PACKED_WEIGHTS): staging is a coalesced copy
  int nu[3];               // 32-channel units per l
};

struct WqUnit {
  int l, cb, u0, xbase;
};
__device__ __forceinline__ WqUnit wq_unit(const WqArgs& a, int u) {
  WqUnit w;
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

// the stream class of a unit: multiplier, ranges and steps of the class, its window table
struct WqClass {
  int m, n_ranges, n_steps;
  const int32_t* win;
};
__host__ __device__ __forceinline__ int wq_class_steps(int n_ranges, int m) { return ((n_ranges + m - 1) / m + WQ_WAVES - 1) / WQ_WAVES; }
__device__ __forceinline__ WqClass wq_class(const WqArgs& a, int l) {
  WqClass c;
  c.m = (l == 0) ? a.mlong : 1;
  c.n_ranges = (a.n_ranges + c.m - 1) / c.m;
  c.n_steps = wq_class_steps(a.n_ranges, c.m);
  c.win = (l == 0 && a.mlong > 1) ? a.win + 2 * wq_class_steps(a.n_ranges, 1) : a.win;
  return c;
}
// Per-l builds (the other units exit) say the units want different stream lengths (QM9-1024,
// us per launch at 48 / 64 / 80 / 112 / 160 edges per stream: forward l = 0 93 / 87 / 79 / 86 / 81, l = 1 65 / 64 / 76 / 89 / 131, l = 2
// 90 / 109 / 148 / 147 / 157; reverse l = 0 171 / 157 / 145 / 163 / 134, l = 1 102 / 96 / 109 / 96 / 117, l = 2 122 / 113 / 136 / 157 / 190;
// profiles/r06_small_experiments.txt item 6) -- but see below.
// MEASURED AND SWITCHED OFF (round 6, profiles/r06_small_experiments.txt item 6): with all seven units in one launch the classes LOSE --
// QM9-1024, whole step, one box: one class of 80 edges 1.731 ms | 48 x 3 1.775 | 48 x 2 1.790 | 64 x 2 1.747 | 40 x 3 1.796 | one class of
// 48 1.838.  A unit alone on the chip is a latency measurement (290 workgroups, one or two per CU); seven units together are a throughput
// measurement, and there every additional step (barrier, staging, range prologue) of the l > 0 units costs more than their window misses.
// The mechanism stays behind XEQ_WQ_LONG_MULT (development; results do not depend on it, tests/test_gpu_parity.py) with one class as default.
static int wq_long_mult(int64_t n_edges, int n_ranges) {
  const char* env = getenv("XEQ_WQ_LONG_MULT");
  if (env && atoi(env) >= 1 && atoi(env) <= 8) return atoi(env);
  (void)n_edges;
  (void)n_ranges;
  return 1;
}

// Work of a workgroup: (chunk of consecutive steps, unit).  A step is WQ_WAVES consecutive ranges, one per wave; the
// waves share the unit (its rbf_lin rows, staged once in LDS) and, step by step, the window of gathered node rows.
// Consecutive work items are the units of one chunk (they share its records and index arrays) and are dealt to
// workgroups so that they run on one XCD (blocks b and b + 8 share one under round-robin dispatch: speed only).
__device__ __forceinline__ void wq_decode(const WqArgs& a, int nunits, int& s0, int& s1, int& unit) {
  const int b = blockIdx.x;
  int region, j;   // region and item index inside it
  if (a.regions == 8) {
    region = b & 7;
    j = b >> 3;
  } else {
    region = 0;
    j = b;
  }
  const int c = j / nunits;   // chunk inside the region
  unit = nunits - 1 - (j - c * nunits);   // a chunk's l = 2 unit first, its l = 0 units last
  const int base = region * a.region_steps, rend = min(base + a.region_steps, a.n_steps);
  s0 = s1 = 0;   // padding block of the grid: no steps
  int cc = c;
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    if (cc >= 0 && cc < a.lvl_chunks[v]) {
      const int lend = v < 2 ? base + a.lvl_start[v + 1] : rend;
      s0 = base + a.lvl_start[v] + cc * a.lvl_spw[v];
      s1 = min(s0 + a.lvl_spw[v], min(lend, rend));
    }
    cc -= a.lvl_chunks[v];
  }
}

// The window of a step in LDS: rows [w0, w0 + rows) of the gathered node arrays, restricted to the unit's columns, as
// NSL pieces of 128 bytes per row.  Staged with 16-byte loads (8 lanes per piece), read back per row with 4-byte LDS
// reads: a 64-lane dword load from global memory occupies the CU's texture addresser as long as a 16-byte one
// (16 cycles), so the per-row gathers -- 92 per tile on average -- bound the older forms (MI355X: 1 750 cycles per
// tile and CU measured, against 450 for the MFMAs); out of LDS the same reads cost 2 cycles each.
#ifndef XEQ_WQ_WIN_FLOATS
#define XEQ_WQ_WIN_FLOATS 12288
#endif
constexpr int WQ_WIN_FLOATS = XEQ_WQ_WIN_FLOATS;   // 48 KB per workgroup: two workgroups per CU

// rbf_lin rows of the unit as the B operands, per kind (0: gate_state, 1: gate_edge, 2: scalar message) WQ_WK<KS> floats:
//   [split][lane][4]   bf16 packs of W[row][8 kh + j], j = 0..7 (lane: j = lane & 31 -> channel row, kh = lane >> 5); 3 x 64 x 4
//   [s][lane]          f32 tail: position q = 2 s + kh (wq_tail_k: k = 16 + q, then the bias, then zeros); KS x 64
//   wq_bftail(KS): instead of the f32 tail [arrangement v][lane][4]: bf16 packs of slots 8 kh .. 8 kh + 7 of B1 / B2 / B3 (above); 3 x 64 x 4
template <int KS>
constexpr int WQ_WK = 3 * 64 * 4 + (wq_bftail(KS) ? 3 * 64 * 4 : KS * 64);
template <int KS>
__device__ __forceinline__ void wq_stage_weights(const WqArgs& a, const WqUnit& un, const float* __restrict__ w,
                                                 const float* __restrict__ b, float* wl) {
  const int nkind = un.l == 0 ? 3 : 2, B = a.B;
  if (a.packed_w) {   // the unit's block of the packed copy, already in this layout: 16-byte coalesced loads (round 5)
    const int unit_index = (un.l == 0 ? 0 : (un.l == 1 ? a.nu[0] : a.nu[0] + a.nu[1])) + un.cb;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(w + (size_t)unit_index * (3 * WQ_WK<KS>));
    for (int idx = threadIdx.x; idx < nkind * WQ_WK<KS> / 4; idx += blockDim.x) reinterpret_cast<f32x4*>(wl)[idx] = src[idx];
    return;
  }
  auto row_of = [&](int kind, int ln) { return (kind == 0 ? un.u0 : (kind == 1 ? a.C + un.u0 : 2 * a.C + 32 * un.cb)) + (ln & 31); };
  for (int idx = threadIdx.x; idx < nkind * 3 * 64; idx += blockDim.x) {   // one 16-byte pack per thread and trip
    const int kind = idx / 192, rem = idx - 192 * kind, split = rem >> 6, ln = rem & 63;
    const int row = row_of(kind, ln), kh = ln >> 5;
    uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int k = 8 * kh + jj;
      uint32_t a3[3];
      wq_split3(k < B ? w[(int64_t)row * B + k] : 0.f, a3[0], a3[1], a3[2]);
      pk[jj >> 1] |= a3[split] << (16 * (jj & 1));
    }
    *reinterpret_cast<f32x4*>(wl + kind * WQ_WK<KS> + 4 * (64 * split + ln)) =
        f32x4{__uint_as_float(pk[0]), __uint_as_float(pk[1]), __uint_as_float(pk[2]), __uint_as_float(pk[3])};
  }
  if constexpr (wq_bftail(KS)) {
    for (int idx = threadIdx.x; idx < nkind * 3 * 64; idx += blockDim.x) {   // one 16-byte pack per thread and trip
      const int kind = idx / 192, rem = idx - 192 * kind, v = rem >> 6, ln = rem & 63;
      const int row = row_of(kind, ln), kh = ln >> 5;
      uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int slot = 8 * kh + jj, part = slot / 5, q = slot - 5 * part;   // slot 15: padding
        const int k = wq_tail_k(q, B);
        uint32_t a3[3];
        wq_split3(k >= 0 ? w[(int64_t)row * B + k] : (k == -1 ? b[row] : 0.f), a3[0], a3[1], a3[2]);
        // arrangement v multiplies the record's part `part` (0 hi, 1 mid, 2 lo) by the weight's split v, where part + v <= 2
        const uint32_t val = (slot < 15 && part + v <= 2) ? a3[v] : 0u;
        pk[jj >> 1] |= val << (16 * (jj & 1));
      }
      *reinterpret_cast<f32x4*>(wl + kind * WQ_WK<KS> + 768 + 4 * (64 * v + ln)) =
          f32x4{__uint_as_float(pk[0]), __uint_as_float(pk[1]), __uint_as_float(pk[2]), __uint_as_float(pk[3])};
    }
    return;
  }
  for (int idx = threadIdx.x; idx < nkind * KS * 64; idx += blockDim.x) {
    const int kind = idx / (KS * 64), rem = idx - kind * (KS * 64), sstep = rem >> 6, ln = rem & 63;
    const int row = row_of(kind, ln), k = wq_tail_k(2 * sstep + (ln >> 5), B);
    wl[kind * WQ_WK<KS> + 768 + rem] = k >= 0 ? w[(int64_t)row * B + k] : (k == -1 ? b[row] : 0.f);
  }
}

// what a lane holds of the record of the MFMA row it owns: the three bf16 packs of its eight k and KS tail values
template <int KS>
struct WqR {
  f32x4 b[3];
  float f[wq_bftail(KS) ? 4 : KS];   // wq_bftail: the four dwords of the lane's eight tail slots
};
// W: the kind's block of the staged weights, + 4 lane for the packs / + 768 + lane for the tail (wq_wptr)
template <int KS>
__device__ __forceinline__ f32x16 wq_filter(const WqR<KS>& R, const float* W, int lane) {
  f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f32x4* Wb = reinterpret_cast<const f32x4*>(W) + lane;
  const bf16x8 ah = __builtin_bit_cast(bf16x8, R.b[0]), am = __builtin_bit_cast(bf16x8, R.b[1]), al = __builtin_bit_cast(bf16x8, R.b[2]);
  const bf16x8 bh = __builtin_bit_cast(bf16x8, Wb[0]), bm = __builtin_bit_cast(bf16x8, Wb[64]), bl = __builtin_bit_cast(bf16x8, Wb[128]);
  // small terms first
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, d, 0, 0, 0);
  if constexpr (wq_bftail(KS)) {   // the tail block against its three weight arrangements, smallest first (a_hi b_lo | a (b_mid) | a (b_hi))
    const f32x4 tv = {R.f[0], R.f[1], R.f[2], R.f[3]};
    const bf16x8 at = __builtin_bit_cast(bf16x8, tv);
    const f32x4* Wt = reinterpret_cast<const f32x4*>(W + 768) + lane;
    const bf16x8 t1 = __builtin_bit_cast(bf16x8, Wt[0]), t2 = __builtin_bit_cast(bf16x8, Wt[64]), t3 = __builtin_bit_cast(bf16x8, Wt[128]);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, t3, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, t2, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, t1, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d, 0, 0, 0);
    return d;
  }
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, d, 0, 0, 0);
  const float* Wf = W + 768 + lane;
#pragma unroll
  for (int s = 0; s < KS; ++s) d = __builtin_amdgcn_mfma_f32_32x32x2f32(R.f[s], Wf[s * 64], d, 0, 0, 0);
  return d;
}

__device__ __forceinline__ float wq_ld(const float* __restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void wq_st(float* __restrict__ base, uint32_t byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// The tile table (LDS, private to the wave, double-buffered).  Position p = 16 * half + v is row v of that half's
// stream in this tile (v = accumulator register); quad slot c = 4 * half + g.
//   [T_G0 + p], [T_G1 + p]      byte offsets of the row's two gathered node rows
//   [T_Y + 32 m + p]            Y_lm of the row (m = 0..2: Y_1, 3..7: Y_2)
//   [T_QOWN + c]                owner node of the quad
//   [T_QKEEP + c]               0 where the quad starts a segment (running sums restart), else 1
//   [T_QLAST + c]               1 where the quad ends a segment (the node's sums are stored)
enum { T_G0 = 0, T_G1 = 32, T_Y = 64, T_QOWN = 320, T_QKEEP = 328, T_QLAST = 336, T_SIZE = 344 };

// what a lane loads for the MFMA row it owns: i = lane & 31 -> half hr = (i >> 2) & 1, register v = 4 (i >> 3) + (i & 3)
template <int KS, int NREC, bool WITH_Y>
struct WqRow {
  WqR<KS> R[NREC];
  f32x4 ya, yb;
  int g;
  uint32_t qi;
};
struct WqStreams {
  int q0, q1, q2, ntiles;
};
// Rows beyond the stream's end (`valid` false) read slot 0's record as it stands (round 6; they used to be zeroed, 38 selects per tile
// and lane in the reverse kernel): their quad carries neither the FIRST nor the LAST flag and comes behind the stream's last real quad
// -- whose LAST flag has stored the node's sums --, so whatever such a row adds to the running sums is never stored, and the per-edge
// partials of the reverse pass are stored for slots inside the stream only (`keeper`).
template <int KS>
__device__ __forceinline__ void wq_load_rec(const float* __restrict__ rec, uint32_t ps, int kh, bool valid, WqR<KS>& R) {
  const f32x4* __restrict__ rp = reinterpret_cast<const f32x4*>(rec + (size_t)ps * wq_recf(KS) + 12 * kh);
#pragma unroll
  for (int c = 0; c < 3; ++c) R.b[c] = rp[c];
  const f32x4* __restrict__ tp = reinterpret_cast<const f32x4*>(rec + (size_t)ps * wq_recf(KS) + WQ_TAIL + wq_tailw(KS) * kh);
#pragma unroll
  for (int c = 0; c < wq_tailw(KS) / 4; ++c) {
    const f32x4 tl = tp[c];
#pragma unroll
    for (int s = 4 * c; s < 4 * c + 4; ++s)
      if (s < (wq_bftail(KS) ? 4 : KS)) R.f[s] = tl[s - 4 * c];
  }
}
template <int KS, int NREC, bool WITH_Y>
__device__ __forceinline__ void wq_row(const WqArgs& a, const WqStreams& st, int lane, int t, const float* __restrict__ rec,
                                       const float* __restrict__ drec, WqRow<KS, NREC, WITH_Y>& w, int g_default = 0) {
  const int i = lane & 31, kh = lane >> 5, hr = (i >> 2) & 1, g = i >> 3;
  const int qb = hr ? st.q1 : st.q0, qe = hr ? st.q2 : st.q1;
  const int q = qb + 4 * t + g;
  const bool valid = q < qe;
  // invalid rows (beyond the stream's end) read slot 0 / quad 0 -- always in bounds -- and are then zeroed: no load sits
  // behind a branch, so the waits of the tile loop stay counted
  const uint32_t ps = valid ? 4u * (uint32_t)q + (uint32_t)(i & 3) : 0u;
  const int gv = a.pgath[ps];
  const uint32_t qv = a.qinfo[valid ? q : 0];
  w.g = valid ? gv : g_default;
  w.qi = valid ? qv : 0u;
  wq_load_rec<KS>(rec, ps, kh, valid, w.R[0]);
  if constexpr (NREC > 1) wq_load_rec<KS>(drec, ps, kh, valid, w.R[1]);
  if constexpr (WITH_Y) {
    const float* __restrict__ ysrc = NREC > 1 ? drec : rec;   // (both records carry Y)
    const f32x4* __restrict__ yp = reinterpret_cast<const f32x4*>(ysrc + (size_t)ps * wq_recf(KS) + wq_yoff(KS));
    w.ya = yp[0];
    w.yb = yp[1];
    if (a.mirror) {   // the mirror edge's vector is the negative of the slot's: d and Y_2 are the same bits, Y_1 changes sign
      w.ya[0] = -w.ya[0];
      w.ya[1] = -w.ya[1];
      w.ya[2] = -w.ya[2];
    }
  }
}
template <int KS, int NREC, bool WITH_Y>
__device__ __forceinline__ void wq_publish(int lane, const WqRow<KS, NREC, WITH_Y>& w, uint32_t stride0, uint32_t stride1,
                                           int* tbl, uint32_t gbase = 0u) {
  // EVERY lane writes, without a branch (round 6).  Lanes l and l + 32 hold the same row (wq_row: i = lane & 31) and the four lanes of
  // a quad the same quad record, so the duplicates store equal values to equal addresses.  Why: behind `if (lane < 32)` / `if ((i & 3)
  // == 0)` the compiler SANK the index loads of wq_row (pgath, qinfo: requested a tile ahead in the source) into the branches that
  // alone use them -- two dependent global round trips per tile, each behind an s_waitcnt vmcnt(0) that also drained the record
  // prefetch (the forward kernel's l = 0 waves: 19.5 % of their cycles waiting for the record and 3.7 % at the publish by the stamps).
  const int i = lane & 31, hr = (i >> 2) & 1, v = 4 * (i >> 3) + (i & 3), p = 16 * hr + v;
#ifdef XEQ_WQ_ABLATE_GATHER   // development: every gather reads node 0 (cache-resident): what the gathers cost
  tbl[T_G0 + p] = 0;
  tbl[T_G1 + p] = 0;
#else
  tbl[T_G0 + p] = (int)(((uint32_t)w.g - gbase) * stride0);   // window mode: gbase = first row of the window
  tbl[T_G1 + p] = (int)(((uint32_t)w.g - gbase) * stride1);
#endif
  if constexpr (WITH_Y) {
    float* tf = reinterpret_cast<float*>(tbl);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tf[T_Y + 32 * q + p] = w.ya[q];
      tf[T_Y + 32 * (4 + q) + p] = w.yb[q];
    }
  }
  const int c = 4 * hr + (i >> 3);
  tbl[T_QOWN + c] = (int)(w.qi & WQ_OWNER);
  tbl[T_QKEEP + c] = (w.qi & WQ_FIRST) ? 0 : 1;
  tbl[T_QLAST + c] = (w.qi & WQ_LAST) ? 1 : 0;
}
template <typename T>
__device__ __forceinline__ void wq_tread4(const int* tbl, int idx, T (&out)[4]) {
  static_assert(sizeof(T) == 4, "table entries are dwords");
  typedef T vec4 __attribute__((ext_vector_type(4)));
  const vec4 v = *reinterpret_cast<const vec4*>(reinterpret_cast<const T*>(tbl) + idx);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = v[i];
}

// per-lane byte columns of the unit
struct WqCols {
  uint32_t b_hs, b_hm, b_s, b_xe, b_x, xnode_b, xcomp_b;
  int64_t x_base;
};
template <int NM>
__device__ __forceinline__ WqCols wq_cols(const WqArgs& a, const WqUnit& un, int j) {
  WqCols w;
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
__device__ __forceinline__ WqStreams wq_streams(const WqArgs& a, int range, int m = 1) {
  WqStreams s;
  const int last = 2 * a.n_ranges;   // (a class stream is m consecutive table streams; the last range of a class may be short)
  s.q0 = a.sq[min(2 * range * m, last)];
  s.q1 = a.sq[min((2 * range + 1) * m, last)];
  s.q2 = a.sq[min((2 * range + 2) * m, last)];
  const int l0 = s.q1 - s.q0, l1 = s.q2 - s.q1;
  s.ntiles = ((l0 > l1 ? l0 : l1) + 3) >> 2;
  return s;
}
// owner nodes of the wave's two streams that have no edge: fn(node) is called by all 64 lanes
// (nodes without an edge own a quad of padding slots since round 3 -- QuadCount above -- and take the ordinary path; nothing to do here)
template <typename Fn>
__device__ __forceinline__ void wq_for_isolated(const WqArgs&, int, int, Fn) {}

// scheduling fence between the passes of a tile (dev switch: -D'XEQ_WQ_SB()=' compiles them out)
#ifndef XEQ_WQ_SB
#define XEQ_WQ_SB() __builtin_amdgcn_sched_barrier(0)
#endif
// fence between the phases of a forward tile (dev switch: -D'XEQ_WQ_FSB()=' lets the compiler interleave them).  FENCED (template
// parameter of the forward body): the first-block kernel runs WITHOUT the fences since round 6 -- its launch 113.6 -> 109.7 us, the general
// kernel's the same with and without (profiles/r06_small_experiments.txt item 12)
#ifndef XEQ_WQ_FSB
#define XEQ_WQ_FSB() do { if constexpr (FENCED) XEQ_WQ_SB(); } while (0)
#endif
// fences between the quads / passes of a REVERSE tile: none by default since round 3 (this file's reverse half is built with the
// default, register-pressure-aware machine scheduler: with the owner rows fetched at the tile top it orders the loads of a tile
// better than the fenced max-ILP stream did -- reverse launch 404 -> 355 us on QM9-1024; -DXEQ_WQ_RSB_ON restores the fences)
#ifdef XEQ_WQ_RSB_ON
#define XEQ_WQ_RSB() __builtin_amdgcn_sched_barrier(0)
#else
#define XEQ_WQ_RSB() \
  do {               \
  } while (0)
#endif

// development (-DXEQ_WQ_STAMPS): cycles of the l = 0 waves per phase of the forward kernel, summed over a launch
static __device__ unsigned long long g_wq_stamps[32];   // (static: each half of the file has its own)
static __device__ unsigned long long g_wq_wg[8192 * 4];   // -DXEQ_WQ_ROLE_TIME / _FWD: per workgroup of the reverse / forward kernel (hw id | l << 32, xcc id, start, end in 100 MHz ticks)
#ifdef XEQ_WQ_STAMPS
#define WQ_STAMP(i)                                                                                          \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long now_;                                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    st_[i] += now_ - last_;                                                                                  \
    last_ = now_;                                                                                            \
  } while (0)
#else
#define WQ_STAMP(i) \
  do {              \
  } while (0)
#endif

__device__ __forceinline__ float wq_lds(const float* win, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(win) + byte_off);
}
#ifndef XEQ_WQ_FWD_WPE
#define XEQ_WQ_FWD_WPE 2
#endif
#ifndef XEQ_WQ_BWD_WPE
#define XEQ_WQ_BWD_WPE 2
#endif

#ifndef XEQ_WQ_PART_BWD
// ------------------------------------------------------------------------------------------------ forward

// window of the forward pass: per row the unit's pieces of h (gate_state | gate_edge | scalar message for l = 0) and of
// xhat (NM pieces: component m in BT layout; the contiguous run of 32 NM floats in e3nn layout)
// XZ (l > 0 only): xhat is zero on the unit's columns (first block of the model), so only the gate_edge piece is staged
template <int NM, bool XZ>
__device__ __forceinline__ void wq_stage_fwd(const WqArgs& a, const WqUnit& un, const WqCols& wc, const float* __restrict__ h,
                                             const float* __restrict__ xhat_, int w0, int nrows, float* win) {
  constexpr int NH = NM == 1 ? 3 : 2, NSL = NH + NM;
  if constexpr (XZ && NM > 1) {
    const int total = nrows * 8;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int row = idx >> 3, chunk = idx & 7;
      *reinterpret_cast<f32x4*>(win + (row * NSL + 1) * 32 + 4 * chunk) =
          *reinterpret_cast<const f32x4*>(h + (int64_t)(w0 + row) * a.H + a.C + un.u0 + 4 * chunk);
    }
    return;
  }
  const int total = nrows * NSL * 8;   // 16-byte chunks
  const int64_t xnode = wc.xnode_b / 4, xcomp = a.xl == 0 ? 32 : wc.xcomp_b / 4;
  constexpr int UN = 8;                // loads in flight per thread before the first LDS store
  const int nthr = blockDim.x;
  for (int base = threadIdx.x; base < total; base += UN * nthr) {
    f32x4 v[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int idx = base + k * nthr;
      const int piece = idx >> 3, chunk = idx & 7;
      const int row = piece / NSL, sl = piece - row * NSL;
      const int64_t n = w0 + (idx < total ? row : 0);
      const float* src = sl < NH ? h + n * a.H + (sl == 0 ? un.u0 : (sl == 1 ? a.C + un.u0 : 2 * a.C + 32 * un.cb))
                                 : xhat_ + wc.x_base + n * xnode + (sl - NH) * xcomp;
      v[k] = *reinterpret_cast<const f32x4*>(src + 4 * chunk);   // a slot beyond the window re-reads row w0: in bounds
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int idx = base + k * nthr;
      if (idx < total) *reinterpret_cast<f32x4*>(win + (idx >> 3) * 32 + 4 * (idx & 7)) = v[k];
    }
  }
}

//   x_c += xhat[n] (h_state[n] phi_state) + Y (h_edge[n] phi_edge);   s_c += h_msg[n] phi_msg   (l = 0)
// WIN: the gathered rows of this step are in the LDS window (first node w0); otherwise they are read from global memory
template <int NM, int KS, bool WIN, bool XZ, bool FENCED>
__device__ __forceinline__ void wq_fwd_body(const WqArgs& a, int range, const WqUnit un, const WqCols& wc,
                                            const float* __restrict__ rec, const float* __restrict__ h,
                                            const float* __restrict__ xhat_, const float* __restrict__ s_in,
                                            const float* __restrict__ x_in, const float* wl, float* __restrict__ s_out,
                                            float* __restrict__ x_out, int* tbl, const float* win, int w0,
                                            unsigned long long* st_, unsigned long long& last_) {
  constexpr bool HAS_S = NM == 1;
  constexpr bool NO_STATE = XZ && NM > 1;   // xhat = 0 on these columns: the gate_state term vanishes with its filter and gathers
  constexpr int YOFF = NM == 3 ? 0 : 3;
  constexpr int NH = NM == 1 ? 3 : 2, ROWB = (NH + NM) * 128;
  const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const uint32_t row_s = 4u * (uint32_t)a.F, row_x = 4u * (uint32_t)a.D;
  wq_for_isolated(a, range, lane, [&](int m) {   // s_out = s_in, x_out = x_in on the unit's columns
    if (hh == 0) {
      if constexpr (HAS_S) wq_st(s_out, (uint32_t)m * row_s + wc.b_s, wq_ld(s_in, (uint32_t)m * row_s + wc.b_s));
#pragma unroll
      for (int mm = 0; mm < NM; ++mm)
        wq_st(x_out, (uint32_t)m * row_x + wc.b_xe + 4u * mm, wq_ld(x_in, (uint32_t)m * row_x + wc.b_xe + 4u * mm));
    }
  });
  const WqStreams st = wq_streams(a, range, un.l == 0 ? a.mlong : 1);
  if (st.ntiles == 0) return;
  const float* __restrict__ h_e = h + a.C;
  const float* __restrict__ h_m = h + (2 * a.C + 32 * un.cb - un.u0);
  const float* xhat_m[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) xhat_m[m] = xhat_ + wc.x_base + (int64_t)m * (wc.xcomp_b / 4);
  const uint32_t stride0 = WIN ? (uint32_t)ROWB : 4u * (uint32_t)a.H, stride1 = WIN ? 0u : wc.xnode_b;
  const uint32_t gbase = WIN ? (uint32_t)w0 : 0u;
  // window mode: byte offset of this lane's element of component m inside a row's xhat pieces
  uint32_t lxm[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) lxm[m] = (uint32_t)(NH * 128) + (a.xl == 0 ? 4u * (uint32_t)(j * NM + m) : (uint32_t)(128 * m + 4 * j));
  const float* Ws = wl;
  const float* We = wl + WQ_WK<KS>;
  const float* Wm = wl + 2 * WQ_WK<KS>;

  float acc_s = 0.f, acc_x[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) acc_x[m] = 0.f;

  using Row = WqRow<KS, 1, (NM > 1)>;
  Row row;
  wq_row<KS, 1, (NM > 1)>(a, st, lane, 0, rec, nullptr, row, (int)gbase);
  wq_publish<KS, 1, (NM > 1)>(lane, row, stride0, stride1, tbl, gbase);
  __builtin_amdgcn_wave_barrier();

  // A tile runs as four phases with every load of the tile issued up front (two waves per SIMD: 256 registers), so a
  // wave has ONE dependent wait per phase instead of one per quad (measured before: 52 % of a wave's cycles in
  // s_waitcnt, 28 % in issue stalls behind its own MFMA chain, 12 k cycles per tile):
  //   A  table entries of all 16 rows, the four owners' residual rows, the next tile's record (global, long latency)
  //   B  the gathered rows of all 16 rows (LDS window, or global memory)
  //   C  the MFMA chains of the filter
  //   D  row arithmetic, quad by quad; the finished nodes' stores
  // development (-DXEQ_WQ_DEFER_STORES): a finished node's sums stored at the TOP of the next tile, in front of that tile's requests (l = 0,
  // 1), so that the wait for the record prefetch does not have to drain a store that may or may not have been issued.  The same idea pays
  // in the reverse kernel (its per-edge partials); here it measured 278 -> 284 us per forward pair: off.
#ifdef XEQ_WQ_DEFER_STORES
  constexpr bool DEFER_ST = NM <= 3;
#else
  constexpr bool DEFER_ST = false;
#endif
  float pend_x[DEFER_ST ? 4 : 1][NM], pend_s[DEFER_ST ? 4 : 1];
  uint32_t pend_ob[DEFER_ST ? 4 : 1], pend_os[DEFER_ST ? 4 : 1];
  int pend_last[DEFER_ST ? 4 : 1];
  if constexpr (DEFER_ST) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pend_last[g] = 0;
  }
  auto flush_nodes = [&]() {
    if constexpr (DEFER_ST) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (pend_last[g]) {
#pragma unroll
          for (int m = 0; m < NM; ++m) wq_st(x_out, pend_ob[g] + 4u * m, pend_x[g][m]);
          if constexpr (HAS_S) wq_st(s_out, pend_os[g], pend_s[g]);
        }
    }
  };

  for (int t = 0; t < st.ntiles; ++t) {
    const int* tb = tbl + (t & 1) * T_SIZE;
    int* tnext = tbl + ((t + 1) & 1) * T_SIZE;
    flush_nodes();   // the previous tile's finished nodes
    const WqR<KS> R = row.R[0];
    WQ_STAMP(4);   // waiting for the tile's record
    // ---- phase A
    uint32_t g0[16], g1[WIN ? 1 : 16], ob[4], os[4];
    int keep[4], last[4];
    float res_x[4][NM], res_s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t t0[4];
      wq_tread4<uint32_t>(tb, T_G0 + 16 * hh + 4 * g, t0);
#pragma unroll
      for (int r = 0; r < 4; ++r) g0[4 * g + r] = t0[r];
      if constexpr (!WIN) {
        uint32_t t1[4];
        wq_tread4<uint32_t>(tb, T_G1 + 16 * hh + 4 * g, t1);
#pragma unroll
        for (int r = 0; r < 4; ++r) g1[4 * g + r] = t1[r];
      }
      const uint32_t own = (uint32_t)tb[T_QOWN + 4 * hh + g];
      keep[g] = tb[T_QKEEP + 4 * hh + g];
      last[g] = tb[T_QLAST + 4 * hh + g];
      ob[g] = own * row_x + wc.b_xe;
      os[g] = own * row_s + wc.b_s;
#pragma unroll
      for (int m = 0; m < NM; ++m) res_x[g][m] = wq_ld(x_in, ob[g] + 4u * m);   // the owner's residual row, read per quad
      res_s[g] = HAS_S ? wq_ld(s_in, os[g]) : 0.f;
    }
    XEQ_WQ_FSB();   // the residual rows are requested IN FRONT of the next tile's record: memory returns in order, and the rows are consumed
                    // in this tile (waiting for them must not wait for the record prefetch behind them)
    wq_row<KS, 1, (NM > 1)>(a, st, lane, t + 1, rec, nullptr, row, (int)gbase);   // next tile's record flies under this tile
    XEQ_WQ_FSB();
    WQ_STAMP(5);   // phase A issued
    // ---- phases B, C, D; l = 2 gathers and consumes its rows in two halves of 8 (register budget)
#ifdef XEQ_WQ_FWD_GR
    constexpr int GR = NM == 5 && XEQ_WQ_FWD_GR > 4 ? XEQ_WQ_FWD_GR / 2 : XEQ_WQ_FWD_GR;
#else
    constexpr int GR = NM == 5 ? 8 : 16;
#endif
    f32x16 ds, de, dm;
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += GR) {
      float hs[GR], he[GR], hm[HAS_S ? GR : 1], xv[GR][NM];
#pragma unroll
      for (int u = 0; u < GR; ++u) {
        const int v = r0 + u;
        if constexpr (WIN) {
          const uint32_t oh = g0[v] + 4u * (uint32_t)j;
          if constexpr (!NO_STATE) hs[u] = wq_lds(win, oh);
          he[u] = wq_lds(win, oh + 128u);
          if constexpr (HAS_S) hm[u] = wq_lds(win, oh + 256u);
          if constexpr (!NO_STATE) {
#pragma unroll
            for (int m = 0; m < NM; ++m) xv[u][m] = wq_lds(win, g0[v] + lxm[m]);
          }
        } else {
          const uint32_t oh = g0[v] + wc.b_hs, ox = g1[v] + wc.b_x;
          if constexpr (!NO_STATE) hs[u] = wq_ld(h, oh);
          he[u] = wq_ld(h_e, oh);
          if constexpr (HAS_S) hm[u] = wq_ld(h_m, oh);
          if constexpr (!NO_STATE) {
#pragma unroll
            for (int m = 0; m < NM; ++m) xv[u][m] = wq_ld(xhat_m[m], ox);
          }
        }
      }
      XEQ_WQ_FSB();
      WQ_STAMP(6);   // phase B issued
      if (r0 == 0) {   // ---- phase C
        de = wq_filter<KS>(R, We, lane);
        ds = de;
        if constexpr (!NO_STATE) ds = wq_filter<KS>(R, Ws, lane);
        dm = ds;
        if constexpr (HAS_S) dm = wq_filter<KS>(R, Wm, lane);
        XEQ_WQ_FSB();
        WQ_STAMP(7);   // MFMAs issued
      }
      // ---- phase D
#pragma unroll
      for (int g = r0 / 4; g < (r0 + GR) / 4; ++g) {
        float Y[NM][4];
        if constexpr (NM > 1) {
#pragma unroll
          for (int m = 0; m < NM; ++m) wq_tread4<float>(tb, T_Y + 32 * (YOFF + m) + 16 * hh + 4 * g, Y[m]);
        }
        float xq[NM], sq = 0.f;
#pragma unroll
        for (int m = 0; m < NM; ++m) xq[m] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = 4 * g + r, u = v - r0;
          // explicit fma chains (this file is built with -ffp-contract=off): the window and the global instantiation of
          // this body must round alike, or a node's result would depend on which one its step took
          const float ge = he[u] * de[v];
          if constexpr (NO_STATE) {   // fma(0, gs, c) = c: the same bits as the general form on xhat = 0
#pragma unroll
            for (int m = 0; m < NM; ++m) xq[m] = __builtin_fmaf(Y[m][r], ge, xq[m]);
          } else {
            const float gs = hs[u] * ds[v];
#pragma unroll
            for (int m = 0; m < NM; ++m) xq[m] = __builtin_fmaf(xv[u][m], gs, NM > 1 ? __builtin_fmaf(Y[m][r], ge, xq[m]) : xq[m] + ge);
          }
          if constexpr (HAS_S) sq = __builtin_fmaf(hm[u], dm[v], sq);
        }
        // A node's running sums START from its residual row (round 6; up to round 5 the residual was added at the store).  The residual
        // rows are requested at the tile top; used only inside `if (last)` their loads were SUNK into that branch by the compiler -- a
        // global round trip behind an s_waitcnt vmcnt(0) per finished node, which also drained the next tile's record prefetch.  Consumed
        // here by a select they stay where they are requested.  (Order of a node's additions: ((res + q_0) + q_1) + ... in any batch.)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc_x[m] = (keep[g] ? acc_x[m] : res_x[g][m]) + xq[m];
        if constexpr (HAS_S) acc_s = (keep[g] ? acc_s : res_s[g]) + sq;
        if constexpr (DEFER_ST) {   // the node's only store, at the next tile's top
#pragma unroll
          for (int m = 0; m < NM; ++m) pend_x[g][m] = acc_x[m];
          pend_s[g] = acc_s;
          pend_ob[g] = ob[g];
          pend_os[g] = os[g];
          pend_last[g] = last[g];
        } else if (last[g]) {   // the node's only store
#pragma unroll
          for (int m = 0; m < NM; ++m) wq_st(x_out, ob[g] + 4u * m, acc_x[m]);
          if constexpr (HAS_S) wq_st(s_out, os[g], acc_s);
        }
      }
      if constexpr (GR < 16) XEQ_WQ_FSB();
      WQ_STAMP(8);   // phase D: row arithmetic and stores
    }
    wq_publish<KS, 1, (NM > 1)>(lane, row, stride0, stride1, tnext, gbase);
    __builtin_amdgcn_wave_barrier();
    WQ_STAMP(9);   // next table published
  }
  flush_nodes();   // the last tile's
}


// one role of the forward kernel: the workgroup's steps, each with its window staged first when it fits
template <int NM, int KS, bool XZ, bool FENCED>
__device__ __forceinline__ void wq_fwd_role(const WqArgs& a, int s_beg, int s_end, const WqUnit un, const float* __restrict__ rec,
                                            const float* __restrict__ h, const float* __restrict__ xhat,
                                            const float* __restrict__ s_in, const float* __restrict__ x_in, const float* wl,
                                            float* __restrict__ s_out, float* __restrict__ x_out, int* tbl, float* win) {
  constexpr int NH = NM == 1 ? 3 : 2, ROWB = (NH + NM) * 128;
  const WqCols wc = wq_cols<NM>(a, un, threadIdx.x & 31);
  unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
#ifdef XEQ_WQ_STAMPS
  last_ = __builtin_amdgcn_s_memtime();
#endif
  // the workgroup's chunk [s_beg, s_end) counts UNITS of a.mlong short steps = one long step: the steps of this unit's class in it
  const WqClass cl = wq_class(a, un.l);
  const int c_beg = cl.m == a.mlong ? s_beg : s_beg * a.mlong, c_end = min(cl.m == a.mlong ? s_end : s_end * a.mlong, cl.n_steps);
  for (int step = c_beg; step < c_end; ++step) {
    const int w0 = cl.win[2 * step], nrows = cl.win[2 * step + 1];
#ifdef XEQ_WQ_NO_WINDOW
    const bool use_win = false;
#else
    const bool use_win = nrows > 0 && nrows * ROWB <= WQ_WIN_FLOATS * 4;   // workgroup-uniform
#endif
    WQ_STAMP(0);   // step head
    if (use_win) wq_stage_fwd<NM, XZ>(a, un, wc, h, xhat, w0, nrows, win);
    WQ_STAMP(1);   // window staged
    __syncthreads();
    WQ_STAMP(2);   // barrier behind the staging
    const int range = step * WQ_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: the stream bounds become scalar loads)
    if (range < cl.n_ranges) {
      if (use_win) wq_fwd_body<NM, KS, true, XZ, FENCED>(a, range, un, wc, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl, win, w0, st_, last_);
      else wq_fwd_body<NM, KS, false, XZ, FENCED>(a, range, un, wc, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl, win, 0, st_, last_);
    }
    WQ_STAMP(3);   // body prologue (isolated nodes, first record) -- what the tile stamps did not take
    __syncthreads();   // the window and the tile tables are rewritten by the next step
    WQ_STAMP(10);  // barrier behind the step
  }
#ifdef XEQ_WQ_STAMPS
  if (NM == 1 && (threadIdx.x & 63) == 0) {
    for (int i = 0; i < 11; ++i) atomicAdd(&g_wq_stamps[i], st_[i]);
    atomicAdd(&g_wq_stamps[15], 1ull);
  }
#endif
}

// The units' rbf_lin rows in the kernels' LDS layout, once per weight version (xeq_message_wq_pack_weights): a workgroup per unit runs
// the very staging code of the kernels into global memory.  Why: that staging reads w[row B + k] with one ROW PER LANE -- 64 cache
// lines per wave instruction, 24 such instructions per workgroup -- and sits between every two workgroups of a CU slot: ~11 us of a
// 160-280 us launch (timing-only variant with a coalesced copy: forward 181 -> 170 us, reverse 296 -> 285 us).
template <int KS>
__global__ void __launch_bounds__(256) k_wq_pack_weights(WqArgs a, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out) {
  const WqUnit un = wq_unit(a, (int)blockIdx.x);
  float* dst = out + (size_t)blockIdx.x * (3 * WQ_WK<KS>);
  if (un.l != 0)   // (two kinds only: the third block of the unit's slot reads as zeros)
    for (int idx = threadIdx.x; idx < WQ_WK<KS>; idx += blockDim.x) dst[2 * WQ_WK<KS> + idx] = 0.f;
  wq_stage_weights<KS>(a, un, w, b, dst);
}

// XZ: xhat is zero on every l > 0 column (the model's first message block: XEmbedding hands over x = 0, and the
// equivariant layer norm of zero is zero there); the l = 0 role is the general one
template <int KS, bool XZ>
__global__ void __launch_bounds__(64 * WQ_WAVES) __attribute__((amdgpu_waves_per_eu(XEQ_WQ_FWD_WPE)))
k_message_fwd_wq(WqArgs a, const float* __restrict__ rec, const float* __restrict__ h, const float* __restrict__ xhat,
                 const float* __restrict__ s_in, const float* __restrict__ x_in, const float* __restrict__ w_rbf,
                 const float* __restrict__ b_rbf, float* __restrict__ s_out, float* __restrict__ x_out) {
  __shared__ __attribute__((aligned(16))) float win[WQ_WIN_FLOATS];
  __shared__ __attribute__((aligned(16))) int tbl_all[WQ_WAVES][2 * T_SIZE];
  __shared__ __attribute__((aligned(16))) float wl[3 * WQ_WK<KS>];
  int s_beg, s_end, unit;
  wq_decode(a, a.nu[0] + a.nu[1] + a.nu[2], s_beg, s_end, unit);
  if (s_beg >= s_end) return;   // padding block of the grid / empty chunk of a short region (workgroup-uniform)
  const WqUnit un = wq_unit(a, unit);
  wq_stage_weights<KS>(a, un, w_rbf, b_rbf, wl);
  __syncthreads();
  int* tbl = tbl_all[threadIdx.x >> 6];
#ifdef XEQ_WQ_ONLY_L   // development: register budget of one role
  if (un.l == XEQ_WQ_ONLY_L) wq_fwd_role<2 * XEQ_WQ_ONLY_L + 1, KS, XZ, !XZ>(a, s_beg, s_end, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl, win);
  return;
#endif
#ifdef XEQ_WQ_ROLE_TIME_FWD   // development: where and when every workgroup of the production body ran
  unsigned long long rr0_;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr0_)::"memory");
#endif
  if (un.l == 0) wq_fwd_role<1, KS, false, !XZ>(a, s_beg, s_end, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl, win);
  else if (un.l == 1) wq_fwd_role<3, KS, XZ, !XZ>(a, s_beg, s_end, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl, win);
  else wq_fwd_role<5, KS, XZ, !XZ>(a, s_beg, s_end, un, rec, h, xhat, s_in, x_in, wl, s_out, x_out, tbl, win);
#ifdef XEQ_WQ_ROLE_TIME_FWD
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned long long rr1_;
    unsigned hw, xcc;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(rr1_), "=s"(hw), "=s"(xcc)::"memory");
    g_wq_wg[4 * blockIdx.x + 0] = hw | ((unsigned long long)un.l << 32);
    g_wq_wg[4 * blockIdx.x + 1] = xcc;
    g_wq_wg[4 * blockIdx.x + 2] = rr0_;
    g_wq_wg[4 * blockIdx.x + 3] = rr1_;
  }
#endif
}

#endif
#ifdef XEQ_WQ_PART_BWD    // the reverse half: reverse kernel, edge gradients
// ------------------------------------------------------------------------------------------------ reverse
struct WqParts {
  float* pd;   // [NU][P]      per-unit partial of dL/dd, by padded slot of the reverse walk
  float* y1;   // [nu1][3][P]  per-unit partial of dL/dY_1m
  float* y2;   // [nu2][5][P]
  int64_t P;   // row length (capacity of the padded order)
};

// Sums over the 32 lanes of each half-wave for the 16 rows of a tile, as a register-halving butterfly: at every step a
// lane keeps half of its registers (chosen by one bit of its lane id) and adds the partner lane's copy of them, so
// 8 + 4 + 2 + 1 = 15 (select, select, DPP add) triples replace sixteen five-step trees.  Staged so that a quad's four
// rows collapse as soon as they exist (wq_red_ab: partners j ^ 1, j ^ 2), and the four quad results of a tile
// collapse at its end (wq_red_cd: partners j ^ 4, j ^ 8, then the other 16-lane row).  Afterwards lane j of a half
// holds the total of row  wq_red_row(j) = 4 (2 b2 + b3) + 2 b0 + b1  (b_k = bit k of j) over the half's 32 lanes.
#define XEQ_WQ_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float wq_red_ab(float r0, float r1, float r2, float r3, bool b0, bool b1) {
  const float a0 = (b0 ? r2 : r0) + XEQ_WQ_DPP(b0 ? r0 : r2, 0xB1);   // quad_perm [1,0,3,2]: lane j ^ 1
  const float a1 = (b0 ? r3 : r1) + XEQ_WQ_DPP(b0 ? r1 : r3, 0xB1);
  return (b1 ? a1 : a0) + XEQ_WQ_DPP(b1 ? a0 : a1, 0x4E);              // quad_perm [2,3,0,1]: lane j ^ 2
}
__device__ __forceinline__ float wq_xor4(float v) {   // lane j ^ 4: quad_perm [3,2,1,0] (j ^ 3) then row_half_mirror (j ^ 7)
  const float t = XEQ_WQ_DPP(v, 0x1B);
  return XEQ_WQ_DPP(t, 0x141);
}
__device__ __forceinline__ float wq_red_cd(float q0, float q1, float q2, float q3, bool b2, bool b3) {
  const float c0 = (b2 ? q2 : q0) + wq_xor4(b2 ? q0 : q2);
  const float c1 = (b2 ? q3 : q1) + wq_xor4(b2 ? q1 : q3);
  float r = (b3 ? c1 : c0) + XEQ_WQ_DPP(b3 ? c0 : c1, 0x128);          // row_ror:8: lane j ^ 8
  r += __shfl_xor(r, 16, 64);                                          // the half's other 16-lane row
  return r;
}
__device__ __forceinline__ int wq_red_row(int j) { return 4 * (((j >> 2) & 1) * 2 + ((j >> 3) & 1)) + 2 * (j & 1) + ((j >> 1) & 1); }

// One role of the reverse pass, in passes of two accumulators (value and d/dd filter of one kind).  Per quad the owner
// (neighbor) node's rows are constants:
//   pass S (state): u_m = sum_r phi_s[r] gx[r][m];  g_hs[n] += <xhat[n], u>;  g_xhat[n][m] += h_s[n] u_m
//                   pd[r] = h_s[n] <xhat[n], gx[r]> phi_s'[r]
//   pass E (edge):  g_he[n] += sum_r phi_e[r] <Y[r], gx[r]>;  pd[r] += h_e[n] <Y[r], gx[r]> phi_e'[r]
//                   dL/dY_m[r] = sum_ch h_e[n] phi_e[r] gx[r][m]
//   pass M (msg):   g_hm[n] += sum_r phi_m[r] gs[r];  pd[r] += h_m[n] gs[r] phi_m'[r]
// window of the reverse pass: per row (center node) the unit's run of grad_x (32 NM floats, e3nn layout) and, for l = 0,
// its 32 floats of grad_s
template <int NM>
__device__ __forceinline__ void wq_stage_bwd(const WqArgs& a, const WqUnit& un, const float* __restrict__ grad_s,
                                             const float* __restrict__ grad_x, int w0, int nrows, float* win) {
  constexpr int NSL = NM + (NM == 1 ? 1 : 0);
  const int total = nrows * NSL * 8;   // 16-byte chunks
  constexpr int UN = 8;                // loads in flight per thread before the first LDS store
  const int nthr = blockDim.x;
  for (int base = threadIdx.x; base < total; base += UN * nthr) {
    f32x4 v[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int idx = base + k * nthr;
      const int piece = idx >> 3, chunk = idx & 7;
      const int row = piece / NSL, sl = piece - row * NSL;
      const int64_t n = w0 + (idx < total ? row : 0);
      const float* src = sl < NM ? grad_x + n * a.D + un.xbase + 32 * sl : grad_s + n * a.F + 32 * un.cb;
      v[k] = *reinterpret_cast<const f32x4*>(src + 4 * chunk);
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int idx = base + k * nthr;
      if (idx < total) *reinterpret_cast<f32x4*>(win + (idx >> 3) * 32 + 4 * (idx & 7)) = v[k];
    }
  }
}

template <int NM, int KS, bool WIN, bool FIRST>
__device__ __forceinline__ void wq_bwd_body(const WqArgs& a, int range, int unit, const WqUnit un, const WqCols& wc,
                                            const float* __restrict__ rec, const float* __restrict__ drec,
                                            const float* __restrict__ h, const float* __restrict__ xhat_,
                                            const float* __restrict__ grad_s, const float* __restrict__ grad_x, const float* wl,
                                            float* __restrict__ grad_h, float* __restrict__ grad_xhat_, const WqParts parts,
                                            int* tbl, const float* win, int w0, unsigned long long* st_,
                                            unsigned long long& last_) {
  constexpr bool HAS_S = NM == 1;
  constexpr int YOFF = NM == 3 ? 0 : 3;
  constexpr int ROWB = (NM + (HAS_S ? 1 : 0)) * 128;
  const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const float* __restrict__ xhat = xhat_ + wc.x_base;
  float* __restrict__ grad_xhat = grad_xhat_ + wc.x_base;
  const uint32_t he_off = 4u * (uint32_t)a.C, row_h = 4u * (uint32_t)a.H;
  // grad_h NULL: only dL/dvec is wanted (first block of a force evaluation).  FIRST knows it at compile time (the value
  // filters, the owners' sums and their stores drop out as dead code) and knows xhat = 0 on the l > 0 columns (pass S,
  // whose every term carries a factor xhat, is not run there)
  const bool node_grads = !FIRST && grad_h != nullptr;
  wq_for_isolated(a, range, lane, [&](int m) {   // nobody's neighbor: zero gradients on the unit's columns
    if (hh == 0 && node_grads) {
      wq_st(grad_h, (uint32_t)m * row_h + wc.b_hs, 0.f);
      wq_st(grad_h, (uint32_t)m * row_h + wc.b_hs + he_off, 0.f);
      if constexpr (HAS_S) wq_st(grad_h, (uint32_t)m * row_h + wc.b_hm, 0.f);
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) wq_st(grad_xhat, (uint32_t)m * wc.xnode_b + wc.b_x + mm * wc.xcomp_b, 0.f);
    }
  });
  const WqStreams st = wq_streams(a, range, un.l == 0 ? a.mlong : 1);
  if (st.ntiles == 0) return;
  // gathered rows: grad_x, grad_s of the center (window mode: one LDS row of ROWB bytes holds both)
  const uint32_t stride0 = WIN ? (uint32_t)ROWB : 4u * (uint32_t)a.D, stride1 = WIN ? (uint32_t)ROWB : 4u * (uint32_t)a.F;
  const uint32_t gbase = WIN ? (uint32_t)w0 : 0u;
  const uint32_t lgx = 4u * (uint32_t)(j * NM), lgs = (uint32_t)(NM * 128 + 4 * j);   // window mode: lane offsets in a row
  const float* Ws = wl;
  const float* We = wl + WQ_WK<KS>;
  const float* Wm = wl + 2 * WQ_WK<KS>;

  float a_hs = 0.f, a_he = 0.f, a_hm = 0.f, a_x[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) a_x[m] = 0.f;
  const int half_beg = hh ? st.q1 : st.q0, half_end = hh ? st.q2 : st.q1;
  const bool b0 = j & 1, b1 = j & 2, b2 = j & 4, b3 = j & 8;
  const int my_r = wq_red_row(j);   // the row of a tile whose channel sums this lane holds after the butterfly

  using Row = WqRow<KS, 2, (NM > 1)>;
  Row row;
  wq_row<KS, 2, (NM > 1)>(a, st, lane, 0, rec, drec, row, (int)gbase);
  wq_publish<KS, 2, (NM > 1)>(lane, row, stride0, stride1, tbl, gbase);
  __builtin_amdgcn_wave_barrier();

  // The owners' rows (h_state, h_edge, h_msg, xhat of the unit's columns) of a tile's four quads.  They come from GLOBAL memory (the
  // owner of a quad is the node whose gradient the wave accumulates: not a row of the window), and up to round 5 they were requested
  // inside the tile that multiplies them -- the l = 0 role in its quad loops, the l > 0 roles at the tile top: the phase stamps of
  // round 6 show the first pass of a tile (which waits for them) at 29 % of a wave's cycles against 6 % for the second.  They are now
  // requested ONE TILE AHEAD (l = 0, 1: 16 / 20 registers; the l = 2 role has no register to spare and keeps the tile-top request): as
  // soon as the next tile's table is published its owners are known, and the loads fly under the loop's back edge, the wait for the
  // next records and the first matrix chain.
#ifdef XEQ_WQ_NO_OWNER_PREFETCH
  constexpr bool OWN_AHEAD = false;
#else
  constexpr bool OWN_AHEAD = NM <= 3;
#endif
  struct Owners {
    float hs[4], he[4], hm[HAS_S ? 4 : 1], x[4][NM];
  };
  auto load_owners = [&](const int* tb_) {
    Owners o;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint32_t own = (uint32_t)tb_[T_QOWN + 4 * hh + g];
      o.he[g] = wq_ld(h, own * row_h + wc.b_hs + he_off);
      if constexpr (HAS_S) o.hm[g] = wq_ld(h, own * row_h + wc.b_hm);
      if constexpr (!FIRST || HAS_S) {
        o.hs[g] = wq_ld(h, own * row_h + wc.b_hs);
#pragma unroll
        for (int m = 0; m < NM; ++m) o.x[g][m] = wq_ld(xhat, own * wc.xnode_b + wc.b_x + m * wc.xcomp_b);
      }
    }
    return o;
  };
  Owners own_next;
  if constexpr (OWN_AHEAD) own_next = load_owners(tbl);
#ifdef XEQ_WQ_NO_ROW_EARLY
  constexpr bool ROW_EARLY = false;
#else
#ifndef XEQ_WQ_ROW_EARLY_MAXNM
#define XEQ_WQ_ROW_EARLY_MAXNM 0   // (measured: no change at 1 or 3 -- reverse launch 227-230 us either way -- so the registers stay free)
#endif
  constexpr bool ROW_EARLY = NM <= XEQ_WQ_ROW_EARLY_MAXNM;   // which roles request the next tile's records at the tile top (below)
#endif

  // The per-edge partials of a tile (dL/dd and dL/dY_lm of its sixteen rows per half) are stored at the TOP of the next tile (round 6).
  // A store inside `if (keeper)` is a memory operation that may or may not have been issued, so the compiler has to wait for the record
  // prefetch -- issued in front of it -- with s_waitcnt vmcnt(0), which waits for the store as well: at the tile's end that was the next
  // tile's first act.  Issued here they are a whole tile old when the next wait counts them.
#ifdef XEQ_WQ_NO_DEFER_PARTS
  constexpr bool DEFER = false;
#else
  constexpr bool DEFER = true;
#endif
  float pend_pd = 0.f, pend_y[NM > 1 ? NM : 1];
  int64_t pend_slot = 0;
  bool pend_keep = false;
  float pend_hm[HAS_S ? 4 : 1];          // l = 0, general form: the scalar-message gradients of the tile's finished owners (pass M)
  uint32_t pend_hm_off[HAS_S ? 4 : 1];
  bool pend_hm_last[HAS_S ? 4 : 1];
  if constexpr (HAS_S) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pend_hm_last[g] = false;
  }
  // the node gradients of passes S and E are stored at the next tile's top as well (l = 0, 1; the l = 2 role has no registers for it), which
  // leaves the rows of a tile without a branch: one scheduling region with the matrix chains (OVL below)
#ifdef XEQ_WQ_NO_DEFER_SE
  constexpr bool DEFER_SE = false;
#else
  constexpr bool DEFER_SE = DEFER && !FIRST && NM <= 3;
#endif
  float pend_hs[DEFER_SE ? 4 : 1], pend_he[DEFER_SE ? 4 : 1], pend_gx[DEFER_SE ? 4 : 1][NM];
  uint32_t pend_own[DEFER_SE ? 4 : 1];
  bool pend_se_last[DEFER_SE ? 4 : 1];
  if constexpr (DEFER_SE) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pend_se_last[g] = false;
  }
  auto flush_parts = [&]() {
    if constexpr (DEFER_SE) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (pend_se_last[g]) {
          wq_st(grad_h, pend_own[g] * row_h + wc.b_hs, pend_hs[g]);
          wq_st(grad_h, pend_own[g] * row_h + wc.b_hs + he_off, pend_he[g]);
          const uint32_t ox = pend_own[g] * wc.xnode_b + wc.b_x;
#pragma unroll
          for (int m = 0; m < NM; ++m) wq_st(grad_xhat, ox + m * wc.xcomp_b, pend_gx[g][m]);
        }
    }
    if constexpr (HAS_S && !FIRST) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (pend_hm_last[g]) wq_st(grad_h, pend_hm_off[g], pend_hm[g]);
    }
    if (pend_keep) {
      parts.pd[(int64_t)unit * parts.P + pend_slot] = pend_pd;
      if constexpr (NM > 1) {
        float* dst = NM == 3 ? parts.y1 : parts.y2;
#pragma unroll
        for (int m = 0; m < NM; ++m) dst[((int64_t)un.cb * NM + m) * parts.P + pend_slot] = pend_y[m];
      }
    }
  };

  for (int t = 0; t < st.ntiles; ++t) {
    const int* tb = tbl + (t & 1) * T_SIZE;
    int* tnext = tbl + ((t + 1) & 1) * T_SIZE;
    if constexpr (DEFER) flush_parts();   // the previous tile's partials
    const WqR<KS> R = row.R[0], Rd = row.R[1];
    WQ_STAMP(4);   // waiting for the tile's records
    // gathered gradient rows of one quad (the center's grad_x, NM components per channel)
    auto load_gx = [&](int g, float (&gxq)[4][NM]) {
      uint32_t g0[4];
      wq_tread4<uint32_t>(tb, T_G0 + 16 * hh + 4 * g, g0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < NM; ++m)
          gxq[r][m] = WIN ? wq_lds(win, g0[r] + lgx + 4u * m) : wq_ld(grad_x, g0[r] + wc.b_xe + 4u * m);
    };
    float gxs[HAS_S ? 4 : 1][4][1];   // l = 0: the tile's grad_x rows stay in registers for passes S and E
    if constexpr (HAS_S) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float tmp[4][NM];
        load_gx(g, tmp);
#pragma unroll
        for (int r = 0; r < 4; ++r) gxs[g][r][0] = tmp[r][0];
      }
    }
    float pd[16];
    // l > 0: the owners' rows (h_state, h_edge, xhat of the unit's columns) of the tile's four quads, fetched HERE, once.  Round 2
    // loaded them quad by quad inside the passes, behind scheduling fences between the quads: a load issued in a quad was waited
    // for in that quad -- twelve exposed global round trips per tile (step timeline of round 3, QM9-1024: a range of the l = 1 /
    // l = 2 units took 41 / 55 us against 28 us for l = 0, whose loads the scheduler hoists by itself; 34 / 43 us now).
    const Owners oq = OWN_AHEAD ? own_next : load_owners(tb);   // (the tile-top request where the rows are not a tile ahead)
    // development (-DXEQ_WQ_ROW_EARLY_MAXNM=1 / 3): the NEXT tile's records and indices requested here, a whole tile ahead of the table
    // that is published from them, instead of behind the last matrix chain (the stamps show 8 % of a wave's cycles at the publish and
    // 6 % at the next tile's top; 34 registers).  Measured on the whole step: no change (profiles/r06_small_experiments.txt item 7).
#ifndef XEQ_WQ_NO_ROW_EARLY
    if constexpr (ROW_EARLY) wq_row<KS, 2, (NM > 1)>(a, st, lane, t + 1, rec, drec, row, (int)gbase);
#endif
    WQ_STAMP(5);   // tile top: gathers issued
    // l > 0: the gathered rows are read one component at a time (four rows x one m), used and dropped: out of the LDS
    // window a re-read costs 2 cycles, while holding a quad's 4 x NM values (next to the filters, pd and the per-quad
    // partial sums) spilled 114 registers in the l = 2 role
    auto gx4 = [&](const uint32_t (&g0)[4], int m, float (&gv)[4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gv[r] = WIN ? wq_lds(win, g0[r] + lgx + 4u * m) : wq_ld(grad_x, g0[r] + wc.b_xe + 4u * m);
    };
    // The NEXT pass's matrix chains are written in front of THIS pass's rows (round 6).  With the rows free of branches (the node
    // gradients of a tile are stored at the next tile's top) a pass's rows and the next pass's chains are one scheduling region, and
    // the compiler places ~7 vector instructions between two matrix instructions instead of issuing eighteen of them back to back: a
    // wave overlaps its own matrix chains with its own row arithmetic instead of waiting for the SIMD's other wave to do so.
    // Reverse pair 432 -> 414 us (profiles/r06_small_experiments.txt item 12).  XEQ_WQ_OVERLAP_MAXNM: which roles (default l = 0, 1).
#ifndef XEQ_WQ_OVERLAP_MAXNM
#define XEQ_WQ_OVERLAP_MAXNM 3
#endif
    constexpr bool OVL = NM <= XEQ_WQ_OVERLAP_MAXNM && (DEFER_SE || (FIRST && HAS_S));
    f32x16 de_h, qe_h, dm_h, qm_h;
    if constexpr (FIRST && NM > 1) {
#pragma unroll
      for (int v = 0; v < 16; ++v) pd[v] = 0.f;
    } else {  // ---- pass S
      const f32x16 ds = wq_filter<KS>(R, Ws, lane), qs = wq_filter<KS>(Rd, Ws, lane);
      if constexpr (OVL) {   // development: the NEXT pass's chains issued in front of this pass's rows (one branch-free region: see OVL)
        de_h = wq_filter<KS>(R, We, lane);
        qe_h = wq_filter<KS>(Rd, We, lane);
      }
      WQ_STAMP(6);   // MFMA issue (all passes)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 4 * hh + g;
        const uint32_t own = (uint32_t)tb[T_QOWN + c];
        const int keep = tb[T_QKEEP + c], last = tb[T_QLAST + c];
        const float o_hs = oq.hs[g];
        float o_x[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) o_x[m] = oq.x[g][m];
        uint32_t g0[4];
        if constexpr (!HAS_S) wq_tread4<uint32_t>(tb, T_G0 + 16 * hh + 4 * g, g0);
        float u[NM], dgs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          float gv[4];
          if constexpr (HAS_S) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = gxs[g][r][0];
          } else {
            gx4(g0, m, gv);
          }
          u[m] = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            u[m] = __builtin_fmaf(ds[4 * g + r], gv[r], u[m]);
            dgs[r] = __builtin_fmaf(o_x[m], gv[r], dgs[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pd[4 * g + r] = (o_hs * dgs[r]) * qs[4 * g + r];
        float hsq = 0.f;
#pragma unroll
        for (int m = 0; m < NM; ++m) hsq = __builtin_fmaf(o_x[m], u[m], hsq);
        a_hs = (keep ? a_hs : 0.f) + hsq;
#pragma unroll
        for (int m = 0; m < NM; ++m) a_x[m] = __builtin_fmaf(o_hs, u[m], keep ? a_x[m] : 0.f);
        if constexpr (DEFER_SE) {
          pend_hs[g] = a_hs;
#pragma unroll
          for (int m = 0; m < NM; ++m) pend_gx[g][m] = a_x[m];
          pend_own[g] = own;
          pend_se_last[g] = last && node_grads;
        } else if (last && node_grads) {
          wq_st(grad_h, own * row_h + wc.b_hs, a_hs);
          const uint32_t ox = own * wc.xnode_b + wc.b_x;
#pragma unroll
          for (int m = 0; m < NM; ++m) wq_st(grad_xhat, ox + m * wc.xcomp_b, a_x[m]);
        }
        if constexpr (NM > 1) XEQ_WQ_RSB();
      }
    }
    XEQ_WQ_RSB();   // accumulator lifetimes of the passes stay disjoint
    WQ_STAMP(7);   // rows of pass S
    const int my_q = half_beg + 4 * t + (my_r >> 2);                   // quad of the row this lane reports
    const bool keeper = j < 16 && my_q < half_end;                     // one 16-lane row per half stores
    const int64_t my_slot = 4 * (int64_t)my_q + (my_r & 3);
    {  // ---- pass E
      const f32x16 de = OVL ? de_h : wq_filter<KS>(R, We, lane), qe = OVL ? qe_h : wq_filter<KS>(Rd, We, lane);
      if constexpr (OVL && HAS_S) {
        dm_h = wq_filter<KS>(R, Wm, lane);
        qm_h = wq_filter<KS>(Rd, Wm, lane);
      }
      WQ_STAMP(6);
      float pq[NM > 1 ? NM : 1][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int p0 = 16 * hh + 4 * g, c = 4 * hh + g;
        const uint32_t own = (uint32_t)tb[T_QOWN + c];
        const int keep = tb[T_QKEEP + c], last = tb[T_QLAST + c];
        const float o_he = oq.he[g];
        uint32_t g0[4];
        if constexpr (!HAS_S) wq_tread4<uint32_t>(tb, T_G0 + p0, g0);
        float dge[4], heq = 0.f;   // dge[r] = <Y[r], gx[r]>, one component at a time
#pragma unroll
        for (int r = 0; r < 4; ++r) dge[r] = NM > 1 ? 0.f : gxs[HAS_S ? g : 0][r][0];
        if constexpr (NM > 1) {
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            float Ym[4], gv[4];
            wq_tread4<float>(tb, T_Y + 32 * (YOFF + m) + p0, Ym);
            gx4(g0, m, gv);
#pragma unroll
            for (int r = 0; r < 4; ++r) dge[r] = __builtin_fmaf(Ym[r], gv[r], dge[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = 4 * g + r;
          heq = __builtin_fmaf(de[v], dge[r], heq);
          pd[v] = __builtin_fmaf(o_he * dge[r], qe[v], pd[v]);
        }
        a_he = (keep ? a_he : 0.f) + heq;
        if constexpr (DEFER_SE) pend_he[g] = a_he;
        else if (last && node_grads) wq_st(grad_h, own * row_h + wc.b_hs + he_off, a_he);
        if constexpr (NM > 1) XEQ_WQ_RSB();
      }
      if constexpr (NM > 1) {
        // dL/dY_lm of every row's edge, in a second walk over the quads: the d/dd filter is dead by now, which is the
        // room the per-quad partial sums need (one walk spilled 48 registers in the l = 2 role)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int p0 = 16 * hh + 4 * g;
          const float o_he = oq.he[g];
          uint32_t g0[4];
          wq_tread4<uint32_t>(tb, T_G0 + p0, g0);
          float wy[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) wy[r] = o_he * de[4 * g + r];
#pragma unroll
          for (int m = 0; m < NM; ++m) {   // first half of the sum over the unit's 32 channels
            float gv[4];
            gx4(g0, m, gv);
            pq[m][g] = wq_red_ab(wy[0] * gv[0], wy[1] * gv[1], wy[2] * gv[2], wy[3] * gv[3], b0, b1);
            if constexpr (NM == 5) XEQ_WQ_RSB();   // one component's butterfly at a time
          }
          XEQ_WQ_RSB();
        }
      }
      if constexpr (NM > 1) {
        float* dst = NM == 3 ? parts.y1 : parts.y2;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const float tot = wq_red_cd(pq[m][0], pq[m][1], pq[m][2], pq[m][3], b2, b3);
          if constexpr (DEFER) pend_y[NM > 1 ? m : 0] = tot;
          else if (keeper) dst[((int64_t)un.cb * NM + m) * parts.P + my_slot] = tot;
        }
      }
    }
    XEQ_WQ_RSB();
    if constexpr (!HAS_S && !ROW_EARLY) wq_row<KS, 2, (NM > 1)>(a, st, lane, t + 1, rec, drec, row, (int)gbase);   // next tile's records
    WQ_STAMP(8);   // rows of pass E (+ dL/dY sums)
    if constexpr (HAS_S) {  // ---- pass M
      float gsv[16];          // the centers' grad_s rows: only this pass reads them; they land under its MFMAs
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t g1[4];
        wq_tread4<uint32_t>(tb, T_G1 + 16 * hh + 4 * g, g1);
#pragma unroll
        for (int r = 0; r < 4; ++r) gsv[4 * g + r] = WIN ? wq_lds(win, g1[r] + lgs) : wq_ld(grad_s, g1[r] + wc.b_s);
      }
      const f32x16 dm = (OVL && HAS_S) ? dm_h : wq_filter<KS>(R, Wm, lane), qm = (OVL && HAS_S) ? qm_h : wq_filter<KS>(Rd, Wm, lane);
      if constexpr (!ROW_EARLY) {
        XEQ_WQ_RSB();
        wq_row<KS, 2, (NM > 1)>(a, st, lane, t + 1, rec, drec, row, (int)gbase);   // last MFMAs issued: next tile's records
        XEQ_WQ_RSB();
      }
      WQ_STAMP(6);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 4 * hh + g;
        const uint32_t own = (uint32_t)tb[T_QOWN + c];
        const int keep = tb[T_QKEEP + c], last = tb[T_QLAST + c];
        const float o_hm = oq.hm[HAS_S ? g : 0];
        float hmq = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = 4 * g + r;
          hmq = __builtin_fmaf(dm[v], gsv[v], hmq);
          pd[v] = __builtin_fmaf(o_hm * gsv[v], qm[v], pd[v]);
        }
        a_hm = (keep ? a_hm : 0.f) + hmq;
        if constexpr (DEFER && !FIRST) {   // (this pass's stores come BEHIND the next tile's record request: to the next tile's top with the partials)
          pend_hm[g] = a_hm;
          pend_hm_off[g] = own * row_h + wc.b_hm;
          pend_hm_last[g] = last && node_grads;
        } else if (last && node_grads) wq_st(grad_h, own * row_h + wc.b_hm, a_hm);
      }
    }
    XEQ_WQ_RSB();
    WQ_STAMP(9);   // rows of pass M
    {  // ---- dL/dd of every row's edge: sum over the unit's 32 channels
      float pq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) pq[g] = wq_red_ab(pd[4 * g], pd[4 * g + 1], pd[4 * g + 2], pd[4 * g + 3], b0, b1);
      const float tot = wq_red_cd(pq[0], pq[1], pq[2], pq[3], b2, b3);
      if constexpr (DEFER) {
        pend_pd = tot;
        pend_slot = my_slot;
        pend_keep = keeper;
      } else if (keeper) parts.pd[(int64_t)unit * parts.P + my_slot] = tot;
    }
    WQ_STAMP(11);  // dL/dd channel sums
    wq_publish<KS, 2, (NM > 1)>(lane, row, stride0, stride1, tnext, gbase);
    __builtin_amdgcn_wave_barrier();
    if constexpr (OWN_AHEAD) own_next = load_owners(tnext);   // the next tile's owners are known: their rows fly under the back edge
    WQ_STAMP(12);  // next table published
  }
  if constexpr (DEFER) flush_parts();   // the last tile's
}

template <int NM, int KS, bool FIRST>
__device__ __forceinline__ void wq_bwd_role(const WqArgs& a, int s_beg, int s_end, int unit, const WqUnit un, const float* __restrict__ rec,
                                            const float* __restrict__ drec, const float* __restrict__ h,
                                            const float* __restrict__ xhat, const float* __restrict__ grad_s,
                                            const float* __restrict__ grad_x, const float* wl, float* __restrict__ grad_h,
                                            float* __restrict__ grad_xhat, const WqParts parts, int* tbl, float* win) {
  constexpr int ROWB = (NM + (NM == 1 ? 1 : 0)) * 128;
  const WqCols wc = wq_cols<NM>(a, un, threadIdx.x & 31);
  unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
#ifdef XEQ_WQ_STAMPS
  last_ = __builtin_amdgcn_s_memtime();
#endif
  const WqClass cl = wq_class(a, un.l);   // (as in the forward role: the chunk counts units, the loop the steps of this unit's class)
  const int c_beg = cl.m == a.mlong ? s_beg : s_beg * a.mlong, c_end = min(cl.m == a.mlong ? s_end : s_end * a.mlong, cl.n_steps);
  for (int step = c_beg; step < c_end; ++step) {
    const int w0 = cl.win[2 * step], nrows = cl.win[2 * step + 1];
#ifdef XEQ_WQ_NO_WINDOW
    const bool use_win = false;
#else
    const bool use_win = nrows > 0 && nrows * ROWB <= WQ_WIN_FLOATS * 4;   // workgroup-uniform
#endif
    WQ_STAMP(0);
    if (use_win) wq_stage_bwd<NM>(a, un, grad_s, grad_x, w0, nrows, win);
    WQ_STAMP(1);
    __syncthreads();
    WQ_STAMP(2);
    const int range = step * WQ_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: the stream bounds become scalar loads)
    if (range < cl.n_ranges) {
      if (use_win) wq_bwd_body<NM, KS, true, FIRST>(a, range, unit, un, wc, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, win, w0, st_, last_);
      else wq_bwd_body<NM, KS, false, FIRST>(a, range, unit, un, wc, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, win, 0, st_, last_);
    }
    WQ_STAMP(3);
    __syncthreads();
    WQ_STAMP(10);
  }
#ifdef XEQ_WQ_STAMPS
#ifndef XEQ_WQ_STAMP_L
#define XEQ_WQ_STAMP_L 0
#endif
  if (NM == 2 * XEQ_WQ_STAMP_L + 1 && (threadIdx.x & 63) == 0) {   // reverse-kernel stamps go to slots 16..31 of the counter array
    for (int i = 0; i < 13; ++i) atomicAdd(&g_wq_stamps[16 + i], st_[i]);
    atomicAdd(&g_wq_stamps[31], 1ull);
  }
#endif
}

// FIRST: the model's first message block in a force evaluation -- no node gradients are wanted (grad_h, grad_xhat NULL)
// and xhat is zero on every l > 0 column
template <int KS, bool FIRST>
__global__ void __launch_bounds__(64 * WQ_WAVES) __attribute__((amdgpu_waves_per_eu(XEQ_WQ_BWD_WPE)))
k_message_bwd_wq(WqArgs a, const float* __restrict__ rec, const float* __restrict__ drec, const float* __restrict__ h,
                 const float* __restrict__ xhat, const float* __restrict__ grad_s, const float* __restrict__ grad_x,
                 const float* __restrict__ w_rbf, const float* __restrict__ b_rbf, float* __restrict__ grad_h,
                 float* __restrict__ grad_xhat, WqParts parts) {
  __shared__ __attribute__((aligned(16))) float win[WQ_WIN_FLOATS];
  __shared__ __attribute__((aligned(16))) int tbl_all[WQ_WAVES][2 * T_SIZE];
  __shared__ __attribute__((aligned(16))) float wl[3 * WQ_WK<KS>];
  int s_beg, s_end, unit;
  wq_decode(a, a.nu[0] + a.nu[1] + a.nu[2], s_beg, s_end, unit);
  if (s_beg >= s_end) return;   // padding block of the grid / empty chunk of a short region (workgroup-uniform)
  const WqUnit un = wq_unit(a, unit);
  wq_stage_weights<KS>(a, un, w_rbf, b_rbf, wl);
  __syncthreads();
  int* tbl = tbl_all[threadIdx.x >> 6];
#ifdef XEQ_WQ_ONLY_L   // development: register budget of one role
  if (un.l == XEQ_WQ_ONLY_L)
    wq_bwd_role<2 * XEQ_WQ_ONLY_L + 1, KS, FIRST>(a, s_beg, s_end, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, win);
  return;
#endif
#ifdef XEQ_WQ_ROLE_TIME   // development: where and when every workgroup of the production body ran
  unsigned long long rr0_;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr0_)::"memory");
#endif
  if (un.l == 0) wq_bwd_role<1, KS, FIRST>(a, s_beg, s_end, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, win);
  else if (un.l == 1) wq_bwd_role<3, KS, FIRST>(a, s_beg, s_end, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, win);
  else wq_bwd_role<5, KS, FIRST>(a, s_beg, s_end, unit, un, rec, drec, h, xhat, grad_s, grad_x, wl, grad_h, grad_xhat, parts, tbl, win);
#ifdef XEQ_WQ_ROLE_TIME
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned long long rr1_;
    unsigned hw, xcc;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(rr1_), "=s"(hw), "=s"(xcc)::"memory");
    g_wq_wg[4 * blockIdx.x + 0] = hw | ((unsigned long long)un.l << 32);
    g_wq_wg[4 * blockIdx.x + 1] = xcc;
    g_wq_wg[4 * blockIdx.x + 2] = rr0_;
    g_wq_wg[4 * blockIdx.x + 3] = rr1_;
  }
#endif
}

// dL/dvec from the per-unit partials (by padded slot of the reverse walk), summed in unit order (deterministic),
// chain rule of A1-A3; one thread per padded slot, pads skipped
__global__ void k_wq_edge_grad(const float* __restrict__ vec, const int32_t* __restrict__ peid, const int32_t* __restrict__ mirror,
                               const int32_t* __restrict__ qptr, int64_t N, int64_t P, int nu, int nu1, int nu2,
                               const float* __restrict__ pd, const float* __restrict__ y1, const float* __restrict__ y2,
                               float* __restrict__ grad_vec) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P || p >= 4 * (int64_t)qptr[N]) return;
  int32_t e = peid[p];
  if (e < 0) return;
  if (mirror) {                // mirror walk: the slot's partials belong to the reverse edge
    const int32_t m = mirror[e];
    if (m < 0) {               // no reverse edge in the list (a periodic list one rounding off symmetric): nobody stands for edge e either
      grad_vec[3 * (int64_t)e] = grad_vec[3 * (int64_t)e + 1] = grad_vec[3 * (int64_t)e + 2] = 0.f;
      return;
    }
    e = m;
  }
  float gd = 0.f, q1[3] = {0.f, 0.f, 0.f}, q2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int u = 0; u < nu; ++u) gd += pd[(int64_t)u * P + p];
  for (int u = 0; u < nu1; ++u)
#pragma unroll
    for (int m = 0; m < 3; ++m) q1[m] += y1[((int64_t)u * 3 + m) * P + p];
  for (int u = 0; u < nu2; ++u)
#pragma unroll
    for (int m = 0; m < 5; ++m) q2[m] += y2[((int64_t)u * 5 + m) * P + p];
  const EdgeGeom<float> g = edge_geom<float>(vec[3 * (int64_t)e], vec[3 * (int64_t)e + 1], vec[3 * (int64_t)e + 2]);
  float out[3];
  edge_grad<float>(g, gd, q1, q2, out);
  grad_vec[3 * (int64_t)e] = out[0];
  grad_vec[3 * (int64_t)e + 1] = out[1];
  grad_vec[3 * (int64_t)e + 2] = out[2];
}

// The same for ALL message blocks of an evaluation at once (round 5): the chain rule A1-A3 is linear in (dL/dd, dL/dY_1, dL/dY_2), so
// the blocks' partials are added first -- per quantity in the order the parts are listed, then over the units -- and the chain rule
// runs once: one launch per evaluation instead of one per block plus the sums of their [E, 3] results (three k_wq_edge_grad and two
// elementwise adds per force evaluation before).
struct WqPartList {
  const float* p[XEQ_WQ_MAX_PART_SETS];
  int n;
};
__global__ void k_wq_edge_grad_sum(const float* __restrict__ vec, const int32_t* __restrict__ peid, const int32_t* __restrict__ mirror,
                                   const int32_t* __restrict__ qptr, int64_t N, int64_t P, int nu, int nu1, int nu2, const WqPartList pl,
                                   float* __restrict__ grad_vec) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P || p >= 4 * (int64_t)qptr[N]) return;
  int32_t e = peid[p];
  if (e < 0) return;
  if (mirror) {
    const int32_t m = mirror[e];
    if (m < 0) {   // no mirror edge in the list (a periodic list one rounding off symmetric): then nobody stands for edge e either
      grad_vec[3 * (int64_t)e] = grad_vec[3 * (int64_t)e + 1] = grad_vec[3 * (int64_t)e + 2] = 0.f;
      return;
    }
    e = m;
  }
  float gd = 0.f, q1[3] = {0.f, 0.f, 0.f}, q2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < pl.n; ++s) {
    const float* pd = pl.p[s];
    const float* y1 = pd + (int64_t)nu * P;
    const float* y2 = y1 + (int64_t)nu1 * 3 * P;
    for (int u = 0; u < nu; ++u) gd += pd[(int64_t)u * P + p];
    for (int u = 0; u < nu1; ++u)
#pragma unroll
      for (int m = 0; m < 3; ++m) q1[m] += y1[((int64_t)u * 3 + m) * P + p];
    for (int u = 0; u < nu2; ++u)
#pragma unroll
      for (int m = 0; m < 5; ++m) q2[m] += y2[((int64_t)u * 5 + m) * P + p];
  }
  const EdgeGeom<float> g = edge_geom<float>(vec[3 * (int64_t)e], vec[3 * (int64_t)e + 1], vec[3 * (int64_t)e + 2]);
  float out[3];
  edge_grad<float>(g, gd, q1, q2, out);
  grad_vec[3 * (int64_t)e] = out[0];
  grad_vec[3 * (int64_t)e + 1] = out[1];
  grad_vec[3 * (int64_t)e + 2] = out[2];
}

#endif
static bool wq_supported(int num_basis, int node_dim, const int32_t mul[3]) {
  return num_basis >= 1 && num_basis <= 31 && mul[0] == node_dim && mul[0] > 0 && mul[0] % 32 == 0 && mul[1] >= 0 &&
         mul[1] % 32 == 0 && mul[2] >= 0 && mul[2] % 32 == 0;
}
// padded slots: at most deg + 3 per node that has an edge, four per node that has none; the capacity every buffer of a plan is sized for
static int64_t wq_pcap(int64_t n_nodes, int64_t n_edges) { return (n_edges + 4 * n_nodes + 3) / 4 * 4; }
// 32-bit byte offsets: rows of h (n_nodes * H * 4) and records (pcap * 128)
static bool wq_fits(int64_t n_nodes, int64_t n_edges, int node_dim, const int32_t mul[3]) {
  const int64_t H = node_dim + 2 * (int64_t)(mul[0] + mul[1] + mul[2]);
  return n_nodes >= 0 && n_edges >= 0 && n_nodes * H * 4 < (1ll << 32) && wq_pcap(n_nodes, n_edges) * (int64_t)WQ_REC_MAX * 4 < (1ll << 32);
}

static int wq_check(const char* who, int64_t n_nodes, int64_t n_edges, int n_ranges, int num_basis, int node_dim,
                    const int32_t mul[3], WqArgs& a) {
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (1ll << 31) && n_nodes < (1ll << 30), "%s: bad sizes", who);
  XEQ_CHECK_ARG(n_ranges >= 1, "%s: the stream table must cover every node (n_ranges >= 1)", who);
  if (!wq_supported(num_basis, node_dim, mul)) {
    xeq::set_error("%s: the wave / quad form needs node_dim == mul[0], multiplicities in multiples of 32 and num_basis <= 23", who);
    return XEQ_ERR_UNSUPPORTED;
  }
  XEQ_CHECK_ARG(wq_fits(n_nodes, n_edges, node_dim, mul), "%s: tensors too large for 32-bit byte offsets (shard the batch)", who);
  for (int l = 0; l < 3; ++l) a.ir.mul[l] = mul[l];
  a.C = a.ir.C();
  a.D = a.ir.D();
  a.F = node_dim;
  a.H = a.F + 2 * a.C;
  a.B = num_basis;
  for (int l = 0; l < 3; ++l) a.nu[l] = mul[l] / 32;
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.pcap = wq_pcap(n_nodes, n_edges);
  a.n_ranges = n_ranges;
  return XEQ_OK;
}

// Steps (WQ_WAVES ranges each) a workgroup walks.  Long chunks sized for ~6 workgroups per CU and unit mix over the first 70 % of a
// region's steps, then chunks a third as long: a launch is 5-6 workgroups per CU, each 100-190 us long (the l = 2 unit of a chunk
// runs 1.8x its l = 0 units), and with equal chunks the CUs finished between 300 and 457 us of a 457 us launch (QM9-1024 reverse
// pass, workgroup timeline); the short chunks fill that ragged end.  Regions keep the XCD mapping: block b runs on XCD b % 8 under
// round-robin dispatch (speed only), so region b % 8 is one XCD's contiguous share of the walk and ITS last chunks are the short ones.
static void wq_geometry(WqArgs& a, int nunits, unsigned& grid) {
  a.mlong = wq_long_mult(a.n_edges, a.n_ranges);
  a.n_steps = wq_class_steps(a.n_ranges, a.mlong);   // chunk units: one long step = mlong short steps
  const int64_t want_chunks = (256 * 6 + nunits - 1) / nunits;
  a.steps_per_wg = (int)((a.n_steps + want_chunks - 1) / want_chunks);
  if (a.steps_per_wg < 1) a.steps_per_wg = 1;
  const char* env = getenv("XEQ_WQ_STEPS_PER_WG");   // development
  if (env && atoi(env) > 0) a.steps_per_wg = atoi(env);
  double frac = 0.8, frac2 = 0.2;
  int div = 3;
  const char* ef = getenv("XEQ_WQ_TAPER_FRAC");       // development
  const char* ef2 = getenv("XEQ_WQ_TAPER_FRAC2");
  const char* ed = getenv("XEQ_WQ_TAPER_DIV");
  if (ef) frac = atof(ef);
  if (ef2) frac2 = atof(ef2);
  if (ed && atoi(ed) > 0) div = atoi(ed);
  const int64_t plain_blocks = (int64_t)((a.n_steps + a.steps_per_wg - 1) / a.steps_per_wg) * nunits;
  a.regions = plain_blocks >= 64 ? 8 : 1;
  a.region_steps = (a.n_steps + a.regions - 1) / a.regions;
  const int spw[3] = {a.steps_per_wg, a.steps_per_wg / div < 1 ? 1 : a.steps_per_wg / div, 1};
  const double share[3] = {frac, frac2, 1.0};
  int start = 0;
  for (int v = 0; v < 3; ++v) {
    a.lvl_spw[v] = spw[v];
    a.lvl_start[v] = start;
    int chunks = v < 2 ? (int)(share[v] * a.region_steps / spw[v]) : (a.region_steps - start + spw[v] - 1) / spw[v];
    if (start + (int64_t)chunks * spw[v] > a.region_steps) chunks = (a.region_steps - start + spw[v] - 1) / spw[v];
    if (chunks < 0) chunks = 0;
    a.lvl_chunks[v] = chunks;
    start += chunks * spw[v];
    if (start > a.region_steps) start = a.region_steps;
  }
  grid = (unsigned)((int64_t)a.regions * (a.lvl_chunks[0] + a.lvl_chunks[1] + a.lvl_chunks[2]) * nunits);
}

}  // namespace xeq

using namespace xeq;

// KS: exact-f32 steps of the filter's tail -- the basis functions from k = 16 on and the bias column, two per step
#define XEQ_WQ_DISPATCH_KS(KERNEL, FLAG, ...)                                                                                \
  do {                                                                                                                       \
    const int ks = wq_ks(num_basis);                                                                                         \
    if (ks <= 1) hipLaunchKernelGGL((KERNEL<1, FLAG>), grid, dim3(64 * WQ_WAVES), 0, (hipStream_t)stream, __VA_ARGS__);      \
    else if (ks <= 3) hipLaunchKernelGGL((KERNEL<3, FLAG>), grid, dim3(64 * WQ_WAVES), 0, (hipStream_t)stream, __VA_ARGS__); \
    else if (ks <= 4) hipLaunchKernelGGL((KERNEL<4, FLAG>), grid, dim3(64 * WQ_WAVES), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<8, FLAG>), grid, dim3(64 * WQ_WAVES), 0, (hipStream_t)stream, __VA_ARGS__);              \
  } while (0)
#define XEQ_WQ_DISPATCH(KERNEL, flag, ...)                   \
  do {                                                       \
    if (flag) XEQ_WQ_DISPATCH_KS(KERNEL, true, __VA_ARGS__); \
    else XEQ_WQ_DISPATCH_KS(KERNEL, false, __VA_ARGS__);     \
  } while (0)

extern "C" {

#ifndef XEQ_WQ_PART_BWD
int xeq_message_wq_supported(int num_basis, int node_dim, const int32_t mul[3]) { return wq_supported(num_basis, node_dim, mul) ? 1 : 0; }

int xeq_message_wq_fits(int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]) {
  return wq_supported(num_basis, node_dim, mul) && n_nodes < (1ll << 30) && wq_fits(n_nodes, n_edges, node_dim, mul) ? 1 : 0;
}

int64_t xeq_message_wq_pcap(int64_t n_nodes, int64_t n_edges) { return wq_pcap(n_nodes, n_edges); }

int xeq_message_wq_waves(void) { return WQ_WAVES; }

int xeq_message_wq_record_floats(void) { return wq_recf(4); }   /* up to 23 basis functions */
int xeq_message_wq_record_floats_for(int num_basis) { return wq_recf(wq_ks(num_basis)); }

int64_t xeq_message_wq_plan_workspace(int64_t n_nodes) {
  size_t temp = 0;
  QuadCount op{nullptr, n_nodes};
  hipcub::CountingInputIterator<int64_t> cnt(0);
  hipcub::TransformInputIterator<int32_t, QuadCount, hipcub::CountingInputIterator<int64_t>> it(cnt, op);
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, temp, it, (int32_t*)nullptr, (int)(n_nodes + 1)) != hipSuccess) return -1;
  return (int64_t)temp + 256;
}

int xeq_message_wq_plan(const int32_t* rowptr, const int32_t* perm, const int64_t* owner, const int64_t* gather,
                        int64_t n_nodes, int64_t n_edges, int n_ranges, void* workspace, int64_t workspace_bytes,
                        int32_t* qptr, int32_t* pgath, int32_t* peid, int32_t* qinfo, int32_t* sq, int32_t* sn, int32_t* win,
                        void* stream) {
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_ranges >= 1 && n_nodes < (1ll << 30) && n_edges < (1ll << 31), "xeq_message_wq_plan: bad sizes");
  const int64_t need = xeq_message_wq_plan_workspace(n_nodes);
  XEQ_CHECK_ARG(need >= 0 && workspace_bytes >= need, "xeq_message_wq_plan: workspace of %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
  QuadCount op{rowptr, n_nodes};
  if (n_nodes <= SCAN_WG_MAX_ITEMS) {   // one workgroup, one launch (the grid-wide scan is two)
    static const bool attr_ok = hipFuncSetAttribute((const void*)k_wq_quad_scan, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)wg_scan_lds_bytes(SCAN_WG_MAX_ITEMS)) == hipSuccess;
    XEQ_CHECK_ARG(attr_ok, "xeq_message_wq_plan: cannot reserve LDS for the quad scan");
    hipLaunchKernelGGL(k_wq_quad_scan, dim3(1), dim3(SCAN_WG_THREADS), wg_scan_lds_bytes(n_nodes), (hipStream_t)stream, op, qptr);
    XEQ_CHECK_LAUNCH("xeq_message_wq_plan (quad scan)");
  } else {
    hipcub::CountingInputIterator<int64_t> cnt(0);
    hipcub::TransformInputIterator<int32_t, QuadCount, hipcub::CountingInputIterator<int64_t>> it(cnt, op);
    size_t temp = (size_t)workspace_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(workspace, temp, it, qptr, (int)(n_nodes + 1), (hipStream_t)stream) != hipSuccess) {
      xeq::set_error("xeq_message_wq_plan: scan failed");
      return XEQ_ERR_LAUNCH;
    }
  }
  if (n_edges + n_nodes > 0) {
    hipLaunchKernelGGL(k_wq_fill, dim3((unsigned)((n_edges + n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_edges, n_nodes,
                       rowptr, perm, owner, gather, (const int32_t*)qptr, pgath, peid, (uint32_t*)qinfo);
    XEQ_CHECK_LAUNCH("xeq_message_wq_plan (fill)");
  }
  const int n = 2 * n_ranges + 1;
  hipLaunchKernelGGL(k_wq_streams, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int32_t*)qptr,
                     n_nodes, n_ranges, sq, sn);
  XEQ_CHECK_LAUNCH("xeq_message_wq_plan (streams)");
  const int n_steps = wq_class_steps(n_ranges, 1);
  hipLaunchKernelGGL(k_wq_windows, dim3((unsigned)n_steps), dim3(64), 0, (hipStream_t)stream, (const int32_t*)sq,
                     (const int32_t*)pgath, n_ranges, n_steps, 1, win);
  XEQ_CHECK_LAUNCH("xeq_message_wq_plan (windows)");
  const int mlong = wq_long_mult(n_edges, n_ranges);
  if (mlong > 1) {   // the long class of the l = 0 units (wq_long_mult): its window table behind the short class's
    const int n_long = wq_class_steps(n_ranges, mlong);
    hipLaunchKernelGGL(k_wq_windows, dim3((unsigned)n_long), dim3(64), 0, (hipStream_t)stream, (const int32_t*)sq,
                       (const int32_t*)pgath, n_ranges, n_long, mlong, win + 2 * n_steps);
    XEQ_CHECK_LAUNCH("xeq_message_wq_plan (windows, long class)");
  }
  return XEQ_OK;
}

int64_t xeq_message_wq_win_ints(int n_ranges) {   /* ints of the plan's window table: both stream classes (include/xeq.h) */
  return 2 * ((int64_t)wq_class_steps(n_ranges, 1) + (int64_t)wq_class_steps(n_ranges, 2) + 1);
}

int xeq_edge_basis_wq(const void* vec, int64_t n_nodes, int64_t n_edges, const int32_t* qptr, const int32_t* peid,
                      int rbf_kind, int cutoff_kind, int num_basis, double cutoff, const void* p0, const void* p1,
                      void* basis, void* dbasis, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && n_nodes >= 0 && num_basis >= 1 && num_basis <= 31 && cutoff > 0, "xeq_edge_basis_wq: bad sizes");
  XEQ_CHECK_ARG(rbf_kind >= XEQ_RBF_BESSEL && rbf_kind <= XEQ_RBF_EXPNORM, "xeq_edge_basis_wq: rbf kernel %d is not implemented", rbf_kind);
  XEQ_CHECK_ARG(rbf_kind == XEQ_RBF_BESSEL || p1 != nullptr, "xeq_edge_basis_wq: this radial basis needs its second parameter array (std / logc / mu)");
  XEQ_CHECK_ARG(cutoff_kind == XEQ_CUTOFF_COSINE || cutoff_kind == XEQ_CUTOFF_POLYNOMIAL, "xeq_edge_basis_wq: cutoff function %d is not implemented", cutoff_kind);
  if (n_edges == 0 && n_nodes == 0) return XEQ_OK;   // (no edge at all: the nodes' lone quads still need their zero records)
  const int64_t pcap = wq_pcap(n_nodes, n_edges), total = (pcap + WQ_REC_SLOTS - 1) / WQ_REC_SLOTS * 256;   // a workgroup per WQ_REC_SLOTS records
  const int ks = wq_ks(num_basis);
  XEQ_CHECK_ARG(pcap * wq_recf(ks) < (1ll << 31) * 2, "xeq_edge_basis_wq: too many edges for 32-bit record offsets (shard the batch)");
  RadialSpec rs{rbf_kind, cutoff_kind, num_basis, cutoff};
  hipLaunchKernelGGL(k_wq_records, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)vec,
                     peid, qptr, n_nodes, pcap, rs, (const float*)p0, (const float*)p1, (float*)basis, (float*)dbasis, wq_recf(ks), wq_tailw(ks), wq_bftail(ks) ? 1 : 0);
  XEQ_CHECK_LAUNCH("xeq_edge_basis_wq");
  return XEQ_OK;
}

static int wq_ks_template(int num_basis) {   // the KS the dispatch macros instantiate for this basis width
  const int ks = wq_ks(num_basis);
  return ks <= 1 ? 1 : (ks <= 3 ? 3 : (ks <= 4 ? 4 : 8));
}
int64_t xeq_message_wq_packed_weight_floats(int num_basis, int node_dim, const int32_t mul[3]) {
  if (!wq_supported(num_basis, node_dim, mul)) return -1;
  const int kst = wq_ks_template(num_basis);
  const int64_t per_unit = 3 * (int64_t)(3 * 64 * 4 + (wq_bftail(kst) ? 3 * 64 * 4 : kst * 64));
  return per_unit * (mul[0] / 32 + mul[1] / 32 + mul[2] / 32);
}
int xeq_message_wq_pack_weights(const void* w_rbf, const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* packed,
                                void* stream) {
  WqArgs a{};
  int rcode = wq_check("xeq_message_wq_pack_weights", 0, 0, 1, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  XEQ_CHECK_ARG(w_rbf && b_rbf && packed, "xeq_message_wq_pack_weights: NULL argument");
  const int nunits = a.nu[0] + a.nu[1] + a.nu[2];
  const dim3 grid((unsigned)nunits);
  switch (wq_ks_template(num_basis)) {
    case 1: hipLaunchKernelGGL(k_wq_pack_weights<1>, grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)w_rbf, (const float*)b_rbf, (float*)packed); break;
    case 3: hipLaunchKernelGGL(k_wq_pack_weights<3>, grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)w_rbf, (const float*)b_rbf, (float*)packed); break;
    case 4: hipLaunchKernelGGL(k_wq_pack_weights<4>, grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)w_rbf, (const float*)b_rbf, (float*)packed); break;
    default: hipLaunchKernelGGL(k_wq_pack_weights<8>, grid, dim3(256), 0, (hipStream_t)stream, a, (const float*)w_rbf, (const float*)b_rbf, (float*)packed); break;
  }
  XEQ_CHECK_LAUNCH("xeq_message_wq_pack_weights");
  return XEQ_OK;
}

int xeq_message_fwd_wq(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* sq, const int32_t* sn, const int32_t* win,
                       const int32_t* c_rowptr, const int32_t* pgath, const int32_t* qinfo, const void* basis, const void* h,
                       const void* xhat, const void* s_in, const void* x_in, const void* w_rbf, const void* b_rbf,
                       int num_basis, int node_dim, const int32_t mul[3], void* s_out, void* x_out, int xhat_layout,
                       void* stream) {
  WqArgs a{};
  int rcode = wq_check("xeq_message_fwd_wq", n_nodes, n_edges, n_ranges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.sq = sq;
  a.sn = sn;
  a.rowptr = c_rowptr;
  a.pgath = pgath;
  a.qinfo = (const uint32_t*)qinfo;
  a.xl = xhat_layout & 1;
  a.packed_w = (xhat_layout & XEQ_WQ_PACKED_WEIGHTS) ? 1 : 0;
  const bool x_zero = (xhat_layout & XEQ_XHAT_HIGHER_L_ZERO) != 0;
  a.win = win;
  const int nunits = a.nu[0] + a.nu[1] + a.nu[2];
  unsigned nblocks;
  wq_geometry(a, nunits, nblocks);
  dim3 grid(nblocks);
  XEQ_WQ_DISPATCH(k_message_fwd_wq, x_zero, a, (const float*)basis, (const float*)h, (const float*)xhat, (const float*)s_in,
                  (const float*)x_in, (const float*)w_rbf, (const float*)b_rbf, (float*)s_out, (float*)x_out);
  XEQ_CHECK_LAUNCH("xeq_message_fwd_wq");
  return XEQ_OK;
}

#endif
#ifdef XEQ_WQ_PART_BWD
int64_t xeq_message_wq_parts_floats(int64_t n_nodes, int64_t n_edges, const int32_t mul[3]) {
  return wq_pcap(n_nodes, n_edges) * (int64_t)(mul[0] / 32 + mul[1] / 32 + mul[2] / 32 + 3 * (mul[1] / 32) + 5 * (mul[2] / 32));
}

int xeq_message_bwd_wq(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* sq, const int32_t* sn, const int32_t* win,
                       const int32_t* n_rowptr, const int32_t* pgath, const int32_t* qinfo, const void* basis,
                       const void* dbasis, const void* h, const void* xhat, const void* grad_s, const void* grad_x,
                       const void* w_rbf, const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* grad_h,
                       void* grad_xhat, void* parts, int xhat_layout, void* stream) {
  WqArgs a{};
  int rcode = wq_check("xeq_message_bwd_wq", n_nodes, n_edges, n_ranges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.sq = sq;
  a.sn = sn;
  a.rowptr = n_rowptr;
  a.pgath = pgath;
  a.qinfo = (const uint32_t*)qinfo;
  a.xl = xhat_layout & 1;
  a.mirror = (xhat_layout & XEQ_WQ_MIRROR_WALK) ? 1 : 0;
  a.packed_w = (xhat_layout & XEQ_WQ_PACKED_WEIGHTS) ? 1 : 0;
  const bool first = (xhat_layout & XEQ_XHAT_HIGHER_L_ZERO) != 0 && grad_h == nullptr;
  const int nunits = a.nu[0] + a.nu[1] + a.nu[2];
  WqParts pr;
  pr.P = a.pcap;
  pr.pd = (float*)parts;
  pr.y1 = pr.pd + (int64_t)nunits * pr.P;
  pr.y2 = pr.y1 + (int64_t)a.nu[1] * 3 * pr.P;
  a.win = win;
  unsigned nblocks;
  wq_geometry(a, nunits, nblocks);
  dim3 grid(nblocks);
  XEQ_WQ_DISPATCH(k_message_bwd_wq, first, a, (const float*)basis, (const float*)dbasis, (const float*)h, (const float*)xhat,
                  (const float*)grad_s, (const float*)grad_x, (const float*)w_rbf, (const float*)b_rbf, (float*)grad_h,
                  (float*)grad_xhat, pr);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_wq");
  return XEQ_OK;
}

#endif
#ifdef XEQ_WQ_PART_BWD
#define XEQ_WQ_DBG(name) name##_bwd
#else
#define XEQ_WQ_DBG(name) name
#endif
/* development: read and clear the phase cycle counters of a -DXEQ_WQ_STAMPS build */
int XEQ_WQ_DBG(xeq_wq_debug_wg)(unsigned long long* out) {   // development (-DXEQ_WQ_ROLE_TIME): 8192 x 4, read and clear
  static unsigned long long zero[8192 * 4];
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wq_wg), sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wq_wg), zero, sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  return XEQ_OK;
}

int XEQ_WQ_DBG(xeq_wq_debug_stamps)(unsigned long long out[32]) {
  unsigned long long zero[32] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wq_stamps), sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wq_stamps), zero, sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  return XEQ_OK;
}

#ifdef XEQ_WQ_PART_BWD
int xeq_message_wq_edge_grad(const void* vec, int64_t n_nodes, int64_t n_edges, const int32_t* qptr, const int32_t* peid,
                             const int32_t* mirror, const int32_t mul[3], const void* parts, void* grad_vec, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && mul[0] % 32 == 0 && mul[1] % 32 == 0 && mul[2] % 32 == 0, "xeq_message_wq_edge_grad: bad sizes");
  if (n_edges == 0) return XEQ_OK;
  const int64_t P = wq_pcap(n_nodes, n_edges);
  const int nu1 = mul[1] / 32, nu2 = mul[2] / 32, nunits = mul[0] / 32 + nu1 + nu2;
  const float* pd = (const float*)parts;
  const float* y1 = pd + (int64_t)nunits * P;
  const float* y2 = y1 + (int64_t)nu1 * 3 * P;
  hipLaunchKernelGGL(k_wq_edge_grad, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)vec,
                     peid, mirror, qptr, n_nodes, P, nunits, nu1, nu2, pd, y1, y2, (float*)grad_vec);
  XEQ_CHECK_LAUNCH("xeq_message_wq_edge_grad");
  return XEQ_OK;
}

int xeq_message_wq_edge_grad_sum(const void* vec, int64_t n_nodes, int64_t n_edges, const int32_t* qptr, const int32_t* peid,
                                 const int32_t* mirror, const int32_t mul[3], int n_sets, const void* const* parts, void* grad_vec,
                                 void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && mul[0] % 32 == 0 && mul[1] % 32 == 0 && mul[2] % 32 == 0, "xeq_message_wq_edge_grad_sum: bad sizes");
  XEQ_CHECK_ARG(n_sets >= 1 && n_sets <= XEQ_WQ_MAX_PART_SETS && parts, "xeq_message_wq_edge_grad_sum: 1 .. %d sets of partials", XEQ_WQ_MAX_PART_SETS);
  if (n_edges == 0) return XEQ_OK;
  const int64_t P = wq_pcap(n_nodes, n_edges);
  const int nu1 = mul[1] / 32, nu2 = mul[2] / 32, nunits = mul[0] / 32 + nu1 + nu2;
  WqPartList pl;
  pl.n = n_sets;
  for (int s = 0; s < XEQ_WQ_MAX_PART_SETS; ++s) pl.p[s] = s < n_sets ? (const float*)parts[s] : nullptr;
  for (int s = 0; s < n_sets; ++s) XEQ_CHECK_ARG(pl.p[s], "xeq_message_wq_edge_grad_sum: parts[%d] is NULL", s);
  hipLaunchKernelGGL(k_wq_edge_grad_sum, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)vec,
                     peid, mirror, qptr, n_nodes, P, nunits, nu1, nu2, pl, (float*)grad_vec);
  XEQ_CHECK_LAUNCH("xeq_message_wq_edge_grad_sum");
  return XEQ_OK;
}

#endif
}  // extern "C"
