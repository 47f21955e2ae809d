// Fused XPaiNN message kernels, "scalar broadcast" form (default path, fp32 and fp64).
// Reference dataflow: nn/xpainn.py:140-159; reverse pass for nn/basic.py:143-159.
//
// Measured on MI355X: the row gathers of this op run at 11-18 TB/s out of L2 when a thread
// keeps several independent loads in flight, so the op is bound by instruction issue and by
// dependent-latency chains, not by memory.  This form therefore has NO barriers and NO LDS in
// the forward pass:
//   * the per-edge quantities every channel needs -- f*rho_k(d), f, Y_lm (and their d/dd
//     companions for the reverse pass) -- are evaluated ONCE per model evaluation by
//     k_edge_basis into a 128-byte record per edge (shared by the three message blocks and
//     both directions) instead of 6 x 20 sin/cos per edge;
//   * a 256-thread workgroup walks one node segment at a time, thread t owns gate channel t
//     and scalar channel t with its three rbf_lin rows in registers (channel on the lane);
//   * the edge index is workgroup-uniform, so the record is fetched with SCALAR loads and
//     feeds the 60 filter FMAs per edge as SGPR operands: no LDS traffic, no VGPRs;
//   * edges are processed U at a time: all 8 U row gathers of a batch are issued before the
//     first use;
//   * the segment sum is a register accumulation in CSR order: no atomics, reproducible.
// The reverse pass walks the CSR over neighbors the same way; per-edge dL/dd and dL/dY_lm
// are reduced over the channels with DPP + one LDS slot per wave, finalised once per segment.
#include "xeq_common.h"

namespace xeq {

// record layout (floats): [0, B) radial terms | BP + {0: envelope, 1..3: Y1, 4..8: Y2}   (forward)
//                         [0, B) d/dd radial  | BP + {0: f', 1..3: unit vector, 4: |r|, 5: 1/|r|}  (reverse)
__host__ __device__ inline int eb_bp(int B) { return (B + 3) & ~3; }
__host__ __device__ inline int eb_width(int B) { return eb_bp(B) + 12; }

template <typename T>
__global__ void k_edge_basis(const T* __restrict__ vec, int64_t E, RadialSpec rs, const T* __restrict__ p0,
                             const T* __restrict__ p1, T* __restrict__ eb, T* __restrict__ ed) {
  const int B = rs.num_basis, BP = eb_bp(B), EW = BP + 12;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * (B + 1)) return;
  const int64_t e = t / (B + 1);
  const int k = (int)(t - e * (B + 1));
  const T rc = (T)rs.cutoff;
  EdgeGeom<T> g = edge_geom<T>(vec[3 * e], vec[3 * e + 1], vec[3 * e + 2]);
  T f, df;
  envelope<T>(rs.cutoff_kind, g.d, rc, f, df);
  if (k < B) {
    T rho, drho;
    radial<T>(rs.rbf_kind, g.d, rc, p0[k], p1 ? p1[k] : T(0), rho, drho, k, B);
    eb[e * EW + k] = f * rho;
    if (ed) ed[e * EW + k] = df * rho + f * drho;
  } else {
    T y1[3], y2[5];
    sph_harm_l12<T>(g, y1, y2);
    T* r = eb + e * EW + BP;
    r[0] = f;
#pragma unroll
    for (int m = 0; m < 3; ++m) r[1 + m] = y1[m];
#pragma unroll
    for (int m = 0; m < 5; ++m) r[4 + m] = y2[m];
    r[9] = r[10] = r[11] = T(0);
    for (int q = B; q < BP; ++q) eb[e * EW + q] = T(0);
    if (ed) {
      T* s = ed + e * EW + BP;
      s[0] = df;
      s[1] = g.x;
      s[2] = g.y;
      s[3] = g.z;
      s[4] = g.d;
      s[5] = g.inv_d;
      for (int q = 6; q < 12; ++q) s[q] = T(0);
      for (int q = B; q < BP; ++q) ed[e * EW + q] = T(0);
    }
  }
}

struct SbArgs {
  int64_t n_nodes, n_edges;
  const int32_t* rowptr;
  const int32_t* perm;
  const int64_t* other;  // fwd: neighbor per edge; bwd: center per edge
  int F, C, D, H, B;
  Irreps ir;
  int xl;     // layout of xhat / grad_xhat
  int chunk;  // consecutive nodes per XCD label
  int y0_zero;  // the l = 0 harmonic is 0 instead of 1: the record stands for a TANGENT of the harmonics (training pass, second order)
  int q_accum;  // k_message_bwd_sbq: add to the per-edge products instead of storing them
  int want_gy;  // k_message_bwd_sbq: form dL/dY_1, dL/dY_2 per edge (eight wave reductions per edge otherwise saved)
  int acc_vec;  // k_message_bwd_sb: dL/dvec is ADDED to what grad_vec holds (XEQ_SB_ACCUM_VEC: the blocks of one evaluation share the buffer)
};

template <typename T, int MAXB>
__device__ __forceinline__ void load_w(const T* __restrict__ w, const T* __restrict__ b, int row, int B, bool valid,
                                       T (&wr)[MAXB], T& br) {
#pragma unroll
  for (int k = 0; k < MAXB; ++k) wr[k] = (valid && k < B) ? w[(int64_t)row * B + k] : T(0);
  br = valid ? b[row] : T(0);
}

// filter value of one channel for one edge: scalar record x register-resident weight row.
// rec[k] is workgroup-uniform (SGPR operands).  Measured on gfx950 (scratch/valu_rate.hip): v_fmac_f32 with an
// SGPR source issues at HALF rate (4.3 cycles per wave64 instruction against 2.3 with VGPR sources), and
// v_pk_fma_f32 costs 4.3 cycles with either -- so fp32 uses packed FMAs (two terms per instruction, even/odd
// partial sums); the compiler's own choice for the scalar loop was v_pk_mul + 2 v_add per two terms.
template <typename T, int MAXB>
__device__ __forceinline__ T filt(const T (&w)[MAXB], T bias, const T* __restrict__ rec, T fe) {
  T acc = bias * fe;
#pragma unroll
  for (int k = 0; k < MAXB; ++k) acc += w[k] * rec[k];
  return acc;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MAXB>
__device__ __forceinline__ float filt(const float (&w)[MAXB], float bias, const float* __restrict__ rec, float fe) {
  static_assert(MAXB % 2 == 0, "packed filter needs an even row length");
  const f32x2* __restrict__ r2 = reinterpret_cast<const f32x2*>(rec);  // records are 16-byte aligned
  f32x2 acc = {bias * fe, 0.f};
#pragma unroll
  for (int k = 0; k < MAXB / 2; ++k) {
    const f32x2 wv = {w[2 * k], w[2 * k + 1]};
    acc = __builtin_elementwise_fma(wv, r2[k], acc);
  }
  return acc.x + acc.y;
}

// Divergence-free addressing: a thread without a channel (t >= C or t >= F) works on a clamped,
// valid channel with zero weights, and the number of components a wave loads is wave-uniform
// (extra components of a lane re-read a valid address and are never stored).  Nothing in the edge
// loop is executed under a per-lane exec mask, so the compiler emits no saveexec/branch pairs.
struct ChanMap {
  int tu, ts;        // clamped gate / scalar channel
  bool has_u, has_s;
  int l, off, nm;    // of channel tu
  int wnm;           // wave-uniform max(nm)
};
__device__ __forceinline__ ChanMap chan_map(const SbArgs& a) {
  ChanMap cm;
  const int t = threadIdx.x;
  cm.has_u = t < a.C;
  cm.has_s = t < a.F;
  cm.tu = min(t, a.C - 1);
  cm.ts = min(t, a.F - 1);
  a.ir.locate(cm.tu, cm.l, cm.off);
  cm.nm = 2 * cm.l + 1;
  const bool any2 = __ballot(cm.l == 2) != 0ull, any1 = __ballot(cm.l == 1) != 0ull;
  cm.wnm = any2 ? 5 : (any1 ? 3 : 1);
  return cm;
}
// Y_lm of this lane's l from the scalar record tail [f, Y1(3), Y2(5)]: y[m], m < 5
template <typename T>
__device__ __forceinline__ void lane_y(const T* __restrict__ tail, int l, T (&y)[5], T y00 = T(1)) {
  const T y10 = tail[1], y11 = tail[2], y12 = tail[3];
  const T y20 = tail[4], y21 = tail[5], y22 = tail[6], y23 = tail[7], y24 = tail[8];
  y[0] = l == 0 ? y00 : (l == 1 ? y10 : y20);
  y[1] = l == 1 ? y11 : y21;
  y[2] = l == 1 ? y12 : y22;
  y[3] = y23;
  y[4] = y24;
}

// WPE = resident waves per SIMD the register allocation must allow.  Measured (QM9-1024, fp32, B = 20): these
// kernels are bound by serial latency chains of one workgroup, so occupancy beats registers: 3 -> 4 waves per SIMD
// (<= 128 VGPRs, U = 2 forward / 1 reverse) took the forward pass from 0.39 to 0.30 ms and the reverse pass from
// 0.72 to 0.57 ms; 5 waves (96 VGPRs) spills the weight rows and is 3-5x slower.
// STAGE (few nodes: an MD-sized system, where the launch is ONE segment's latency chain): the records of a group of <= 64 edges are
// copied to LDS with one round of vector loads before the group's arithmetic, instead of one scalar-memory round trip per edge batch in
// front of its filters; the filters then read their record operands from LDS (broadcast reads).  The same arithmetic, the same bits.
template <typename T>
__device__ __forceinline__ void sb_stage_records(const T* __restrict__ src, int EW, int32_t eid_v, int cnt, T* __restrict__ dst, int lane, int wave) {
  constexpr int LPE = sizeof(T) == 4 ? 16 : 32, EPI = 64 / LPE;   // lanes per edge (16 bytes each), edges per wave instruction
  const int parts = EW * (int)sizeof(T) / 16;
  const int part = lane % LPE;
  for (int j0 = wave * EPI; j0 < cnt; j0 += 4 * EPI) {
    const int j = j0 + lane / LPE;
    const int32_t e = __shfl(eid_v, min(j, cnt - 1), 64);
    if (j < cnt && part < parts)
      reinterpret_cast<float4*>(dst + j * EW)[part] = reinterpret_cast<const float4*>(src + (uint32_t)e * (uint32_t)EW)[part];
  }
}

template <typename T, int MAXB, int U, int WPE, bool STAGE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_message_fwd_sb(SbArgs a, const T* __restrict__ eb, const T* __restrict__ h,
                                                        const T* __restrict__ xhat, const T* __restrict__ s_in,
                                                        const T* __restrict__ x_in, const T* __restrict__ w_rbf,
                                                        const T* __restrict__ b_rbf, T* __restrict__ s_out,
                                                        T* __restrict__ x_out) {
  const int B = a.B, C = a.C, F = a.F, D = a.D, H = a.H;
  const int BP = eb_bp(B), EW = BP + 12;
  const ChanMap cm = chan_map(a);
  const XAddr xa = xaddr(a.ir, a.n_nodes, cm.tu, a.xl);
  int xcomp[5];  // component offsets, clamped to the lane's own components
#pragma unroll
  for (int m = 0; m < 5; ++m) xcomp[m] = min(m, cm.nm - 1) * xa.comp;
  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w<T, MAXB>(w_rbf, b_rbf, cm.tu, B, cm.has_u, ws, bs);
  load_w<T, MAXB>(w_rbf, b_rbf, C + cm.tu, B, cm.has_u, we, be);
  load_w<T, MAXB>(w_rbf, b_rbf, 2 * C + cm.ts, B, cm.has_s, wm, bm);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  __shared__ __attribute__((aligned(16))) T srec[STAGE ? 64 * (MAXB + 12) : 4];
  const T y00 = a.y0_zero ? T(0) : T(1);
  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t c = walk.next(); c >= 0; c = walk.next()) {
    const int32_t e0 = a.rowptr[c], e1 = a.rowptr[c + 1];
    T acc_s = T(0), acc_x[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t pb = e0; pb < e1; pb += 64) {
      // the segment's edge ids / neighbour rows: ONE coalesced vector load per 64 edges, handed out
      // with v_readlane (no dependent scalar-memory chain per edge)
      const int cnt = min(64, e1 - pb);
      const int32_t slot_v = pb + min(lane, cnt - 1);
      const int32_t eid_v = a.perm ? a.perm[slot_v] : slot_v;
      const int32_t nbr_v = (int32_t)a.other[eid_v];
      if (STAGE) {
        __syncthreads();   // the previous group's (or segment's) readers are done
        sb_stage_records<T>(eb, EW, eid_v, cnt, srec, lane, wave);
        __syncthreads();
      }
      for (int j0 = 0; j0 < cnt; j0 += U) {
        uint32_t noff[U], eoff[U];  // 32-bit element offsets (host checks N*H and E*EW < 2^31)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = min(j0 + u, cnt - 1);  // a padded slot repeats the last edge and is skipped below
          noff[u] = (uint32_t)__builtin_amdgcn_readlane(nbr_v, j);
          eoff[u] = (uint32_t)__builtin_amdgcn_readlane(eid_v, j) * (uint32_t)EW;
        }
        T hs[U], he[U], hm[U], xv[U][5];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const T* hn = h + noff[u] * (uint32_t)H;
          hm[u] = hn[2 * C + cm.ts];
          hs[u] = hn[cm.tu];
          he[u] = hn[C + cm.tu];
          const T* xn = xhat + xa.off + (int64_t)noff[u] * xa.node;
          xv[u][0] = xn[0];
          if (cm.wnm >= 3) {  // wave-uniform
            xv[u][1] = xn[xcomp[1]];
            xv[u][2] = xn[xcomp[2]];
          } else {
            xv[u][1] = xv[u][2] = T(0);
          }
          if (cm.wnm >= 5) {
            xv[u][3] = xn[xcomp[3]];
            xv[u][4] = xn[xcomp[4]];
          } else {
            xv[u][3] = xv[u][4] = T(0);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (j0 + u < cnt) {  // uniform
            const T* rec = STAGE ? srec + (j0 + u) * EW : eb + eoff[u];  // workgroup-uniform address: scalar loads (STAGE: LDS broadcast reads)
            const T fe = rec[BP];
            const T ps = filt(ws, bs, rec, fe);
            const T pe = filt(we, be, rec, fe);
            const T pm = filt(wm, bm, rec, fe);
            acc_s += hm[u] * pm;
            const T gs = hs[u] * ps, ge = he[u] * pe;
            T y[5];
            lane_y<T>(rec + BP, cm.l, y, y00);
#pragma unroll
            for (int m = 0; m < 5; ++m) acc_x[m] += xv[u][m] * gs + y[m] * ge;
          }
        }
      }
    }
    if (cm.has_s) s_out[c * F + cm.ts] = (s_in ? s_in[c * F + cm.ts] : T(0)) + acc_s;   // s_in / x_in NULL: the aggregate alone (ops.DiffMessage)
    if (cm.has_u) {
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < cm.nm) x_out[c * D + cm.off + m] = (x_in ? x_in[c * D + cm.off + m] : T(0)) + acc_x[m];
    }
  }
}

// wave reduction helpers: DPP for float, shuffles for double
__device__ __forceinline__ float wave_total(float v) {
#define XEQ_SB_DPP(v, ctrl, rmask) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), ctrl, rmask, 0xF, true)))
  v = XEQ_SB_DPP(v, 0xB1, 0xF);
  v = XEQ_SB_DPP(v, 0x4E, 0xF);
  v = XEQ_SB_DPP(v, 0x141, 0xF);
  v = XEQ_SB_DPP(v, 0x140, 0xF);
  v = XEQ_SB_DPP(v, 0x142, 0xA);
  v = XEQ_SB_DPP(v, 0x143, 0xC);
#undef XEQ_SB_DPP
  return __shfl(v, 63, 64);
}
__device__ __forceinline__ double wave_total(double v) { return wave_sum<double>(v); }

constexpr int SB_RED = 64;  // edges whose reduction slots fit in LDS between two finalisations

template <typename T, int MAXB, int U, int WPE, bool STAGE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_message_bwd_sb(SbArgs a, const T* __restrict__ eb, const T* __restrict__ ed,
                                                        const T* __restrict__ h, const T* __restrict__ xhat,
                                                        const T* __restrict__ grad_s, const T* __restrict__ grad_x,
                                                        const T* __restrict__ w_rbf, const T* __restrict__ b_rbf,
                                                        T* __restrict__ grad_h, T* __restrict__ grad_xhat,
                                                        T* __restrict__ grad_vec) {
  __shared__ T red[SB_RED][4][9];
  __shared__ int32_t red_eid[SB_RED];
  __shared__ __attribute__((aligned(16))) T srec[STAGE ? 2 * SB_RED * (MAXB + 12) : 4];   // STAGE: value | derivative records of the group
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.B, C = a.C, F = a.F, D = a.D, H = a.H;
  const int BP = eb_bp(B), EW = BP + 12;
  const ChanMap cm = chan_map(a);
  const bool is1 = cm.has_u && cm.l == 1, is2 = cm.has_u && cm.l == 2;
  const bool wave_has1 = __ballot(is1) != 0ull, wave_has2 = __ballot(is2) != 0ull;
  const XAddr xa = xaddr(a.ir, a.n_nodes, cm.tu, a.xl);
  int gcomp[5];  // offsets inside the e3nn row of grad_x, clamped to the lane's own components
#pragma unroll
  for (int m = 0; m < 5; ++m) gcomp[m] = cm.off + min(m, cm.nm - 1);
  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w<T, MAXB>(w_rbf, b_rbf, cm.tu, B, cm.has_u, ws, bs);
  load_w<T, MAXB>(w_rbf, b_rbf, C + cm.tu, B, cm.has_u, we, be);
  load_w<T, MAXB>(w_rbf, b_rbf, 2 * C + cm.ts, B, cm.has_s, wm, bm);

  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t n = walk.next(); n >= 0; n = walk.next()) {
    const int32_t e0 = a.rowptr[n], e1 = a.rowptr[n + 1];
    const T hs = cm.has_u ? h[n * H + cm.tu] : T(0), he = cm.has_u ? h[n * H + C + cm.tu] : T(0);
    const T hm = cm.has_s ? h[n * H + 2 * C + cm.ts] : T(0);
    T xh[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) xh[m] = (cm.has_u && m < cm.nm) ? xhat[xa.off + n * xa.node + m * xa.comp] : T(0);
    T acc_hs = T(0), acc_he = T(0), acc_hm = T(0), acc_xh[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t pb = e0; pb < e1; pb += SB_RED) {  // groups of <= SB_RED edges share one finalisation
      const int32_t pe_ = min(pb + SB_RED, e1);
      const int cnt = pe_ - pb;  // <= SB_RED = 64: one vector load of edge ids / centre rows
      const int32_t slot_v = pb + min(lane, cnt - 1);
      const int32_t eid_v = a.perm ? a.perm[slot_v] : slot_v;
      const int32_t ctr_v = (int32_t)a.other[eid_v];
      if (STAGE) {   // (the group's previous readers passed the barrier that follows every finalisation)
        sb_stage_records<T>(eb, EW, eid_v, cnt, srec, lane, __builtin_amdgcn_readfirstlane(wave));
        sb_stage_records<T>(ed, EW, eid_v, cnt, srec + SB_RED * (MAXB + 12), lane, __builtin_amdgcn_readfirstlane(wave));
        __syncthreads();
      }
      // the centre rows of edge slot j of this group: dL/ds_out, dL/dx_out of the lane's channels
      auto gather = [&](int j, T (&gxo)[5], T& dgmo) {
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane(ctr_v, j);
        dgmo = grad_s[ci * (uint32_t)F + cm.ts];
        const T* gr = grad_x + ci * (uint32_t)D;
        gxo[0] = gr[gcomp[0]];
        if (cm.wnm >= 3) {  // wave-uniform
          gxo[1] = gr[gcomp[1]];
          gxo[2] = gr[gcomp[2]];
        } else {
          gxo[1] = gxo[2] = T(0);
        }
        if (cm.wnm >= 5) {
          gxo[3] = gr[gcomp[3]];
          gxo[4] = gr[gcomp[4]];
        } else {
          gxo[3] = gxo[4] = T(0);
        }
      };
      // U = 1 (every shipped instantiation): the NEXT edge's rows are requested before this edge's arithmetic -- the per-edge work (six
      // filters, nine wave reductions) then runs under the gather's round trip instead of behind it.  The arithmetic is the one-edge
      // body itself: same sums, same order.
      T gxn[5], dgmn = T(0);
#pragma unroll
      for (int m = 0; m < 5; ++m) gxn[m] = T(0);
      if (U == 1) gather(0, gxn, dgmn);
      for (int32_t p = pb; p < pe_; p += U) {
        int32_t eid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) eid[u] = __builtin_amdgcn_readlane(eid_v, min(p + u, pe_ - 1) - pb);
        T gx[U][5], dgm[U];
        if (U == 1) {
#pragma unroll
          for (int m = 0; m < 5; ++m) gx[0][m] = gxn[m];
          dgm[0] = dgmn;
          gather(min(p + 1, pe_ - 1) - pb, gxn, dgmn);   // (past the group's last edge: that edge again, never used)
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) gather(min(p + u, pe_ - 1) - pb, gx[u], dgm[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int m = 0; m < 5; ++m)
            if (m >= cm.nm) gx[u][m] = T(0);  // components the lane does not own (v_cndmask, no branch)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (p + u < pe_) {  // uniform
            const T* rec = STAGE ? srec + (p + u - pb) * EW : eb + (uint32_t)eid[u] * (uint32_t)EW;
            const T* rd = STAGE ? srec + SB_RED * (MAXB + 12) + (p + u - pb) * EW : ed + (uint32_t)eid[u] * (uint32_t)EW;
            const T fe = rec[BP], dfe = rd[BP];
            const T ps = filt(ws, bs, rec, fe), pe = filt(we, be, rec, fe), pm = filt(wm, bm, rec, fe);
            const T qs = filt(ws, bs, rd, dfe), qe = filt(we, be, rd, dfe), qm = filt(wm, bm, rd, dfe);
            T y[5];
            lane_y<T>(rec + BP, cm.l, y);
            T dgs = T(0), dge = T(0);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
              dgs += xh[m] * gx[u][m];
              dge += y[m] * gx[u][m];
            }
            const T dg_m = cm.has_s ? dgm[u] : T(0);
            acc_hs += ps * dgs;
            acc_he += pe * dge;
            acc_hm += pm * dg_m;
            const T gate = hs * ps;
#pragma unroll
            for (int m = 0; m < 5; ++m) acc_xh[m] += gate * gx[u][m];
            const T pd = wave_total(hs * dgs * qs + he * dge * qe + hm * dg_m * qm);
            const T gy = he * pe;
            T r1[3] = {T(0), T(0), T(0)}, r2[5] = {T(0), T(0), T(0), T(0), T(0)};
            if (wave_has1) {
#pragma unroll
              for (int m = 0; m < 3; ++m) r1[m] = wave_total(is1 ? gy * gx[u][m] : T(0));
            }
            if (wave_has2) {
#pragma unroll
              for (int m = 0; m < 5; ++m) r2[m] = wave_total(is2 ? gy * gx[u][m] : T(0));
            }
            if (lane == 0) {
              const int j = p + u - pb;
              red[j][wave][0] = pd;
#pragma unroll
              for (int m = 0; m < 3; ++m) red[j][wave][1 + m] = r1[m];
#pragma unroll
              for (int m = 0; m < 5; ++m) red[j][wave][4 + m] = r2[m];
              if (wave == 0) red_eid[j] = eid[u];
            }
          }
        }
      }
      __syncthreads();
      if (t < pe_ - pb) {  // one lane per edge: sum the four waves, chain rule to dL/dvec
        T gd = T(0), q1[3] = {T(0), T(0), T(0)}, q2[5] = {T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          gd += red[t][w][0];
#pragma unroll
          for (int m = 0; m < 3; ++m) q1[m] += red[t][w][1 + m];
#pragma unroll
          for (int m = 0; m < 5; ++m) q2[m] += red[t][w][4 + m];
        }
        const int64_t e = red_eid[t];
        const T* rd = ed + e * EW + BP;
        EdgeGeom<T> g;
        g.x = rd[1];
        g.y = rd[2];
        g.z = rd[3];
        g.d = rd[4];
        g.inv_d = rd[5];
        T out[3];
        edge_grad<T>(g, gd, q1, q2, out);
        if (a.acc_vec) {   // (an edge belongs to one segment: no other workgroup touches these three floats)
          out[0] = grad_vec[3 * e] + out[0];
          out[1] = grad_vec[3 * e + 1] + out[1];
          out[2] = grad_vec[3 * e + 2] + out[2];
        }
        grad_vec[3 * e] = out[0];
        grad_vec[3 * e + 1] = out[1];
        grad_vec[3 * e + 2] = out[2];
      }
      __syncthreads();
    }
    if (cm.has_u) {
      grad_h[n * H + cm.tu] = acc_hs;
      grad_h[n * H + C + cm.tu] = acc_he;
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < cm.nm) grad_xhat[xa.off + n * xa.node + m * xa.comp] = acc_xh[m];
    }
    if (cm.has_s) grad_h[n * H + 2 * C + cm.ts] = acc_hm;
  }
}

// The reverse pass in the form a TRAINING pass differentiates again (nn/training.py, ops.DiffMessage): the message is multilinear in
// (dL/dout, h, xhat | Y, record head, rbf_lin rows), so every second-order term is this kernel or the forward kernel with one operand
// replaced by its tangent.  Same walk as k_message_bwd_sb; instead of folding the per-edge sums into dL/dvec it hands out
//   q[e, c]  = h[nbr(e), c] * P[e, c]   (P: the contraction of dL/dx_out with xhat / Y, or dL/ds_out)  -- the caller forms
//              dL/drecord_head = q W' and dL/dW' = record_head^T q with two library GEMMs, q never outlives the call;
//   gy[e, 8] = dL/dY_1, dL/dY_2 of the edge.
template <typename T, int MAXB, int U, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_message_bwd_sbq(SbArgs a, const T* __restrict__ eb,
                                                        const T* __restrict__ h, const T* __restrict__ xhat,
                                                        const T* __restrict__ grad_s, const T* __restrict__ grad_x,
                                                        const T* __restrict__ w_rbf, const T* __restrict__ b_rbf,
                                                        T* __restrict__ grad_h, T* __restrict__ grad_xhat,
                                                        T* __restrict__ q_out, T* __restrict__ gy_out) {
  __shared__ T red[SB_RED][4][8];
  __shared__ int32_t red_eid[SB_RED];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.B, C = a.C, F = a.F, D = a.D, H = a.H;
  const int BP = eb_bp(B), EW = BP + 12;
  const ChanMap cm = chan_map(a);
  const bool is1 = cm.has_u && cm.l == 1, is2 = cm.has_u && cm.l == 2;
  const bool wave_has1 = __ballot(is1) != 0ull, wave_has2 = __ballot(is2) != 0ull;
  const XAddr xa = xaddr(a.ir, a.n_nodes, cm.tu, a.xl);
  const T y00 = a.y0_zero ? T(0) : T(1);
  int gcomp[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) gcomp[m] = cm.off + min(m, cm.nm - 1);
  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w<T, MAXB>(w_rbf, b_rbf, cm.tu, B, cm.has_u, ws, bs);
  load_w<T, MAXB>(w_rbf, b_rbf, C + cm.tu, B, cm.has_u, we, be);
  load_w<T, MAXB>(w_rbf, b_rbf, 2 * C + cm.ts, B, cm.has_s, wm, bm);

  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t n = walk.next(); n >= 0; n = walk.next()) {
    const int32_t e0 = a.rowptr[n], e1 = a.rowptr[n + 1];
    const T hs = cm.has_u ? h[n * H + cm.tu] : T(0), he = cm.has_u ? h[n * H + C + cm.tu] : T(0);
    const T hm = cm.has_s ? h[n * H + 2 * C + cm.ts] : T(0);
    T xh[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) xh[m] = (cm.has_u && m < cm.nm) ? xhat[xa.off + n * xa.node + m * xa.comp] : T(0);
    T acc_hs = T(0), acc_he = T(0), acc_hm = T(0), acc_xh[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t pb = e0; pb < e1; pb += SB_RED) {
      const int32_t pe_ = min(pb + SB_RED, e1);
      const int cnt = pe_ - pb;
      const int32_t slot_v = pb + min(lane, cnt - 1);
      const int32_t eid_v = a.perm ? a.perm[slot_v] : slot_v;
      const int32_t ctr_v = (int32_t)a.other[eid_v];
      for (int32_t p = pb; p < pe_; p += U) {
        int32_t eid[U];
        uint32_t cidx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = min(p + u, pe_ - 1) - pb;
          eid[u] = __builtin_amdgcn_readlane(eid_v, j);
          cidx[u] = (uint32_t)__builtin_amdgcn_readlane(ctr_v, j);
        }
        T gx[U][5], dgm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          dgm[u] = grad_s[cidx[u] * (uint32_t)F + cm.ts];
          const T* gr = grad_x + cidx[u] * (uint32_t)D;
          gx[u][0] = gr[gcomp[0]];
          if (cm.wnm >= 3) {  // wave-uniform
            gx[u][1] = gr[gcomp[1]];
            gx[u][2] = gr[gcomp[2]];
          } else {
            gx[u][1] = gx[u][2] = T(0);
          }
          if (cm.wnm >= 5) {
            gx[u][3] = gr[gcomp[3]];
            gx[u][4] = gr[gcomp[4]];
          } else {
            gx[u][3] = gx[u][4] = T(0);
          }
#pragma unroll
          for (int m = 0; m < 5; ++m)
            if (m >= cm.nm) gx[u][m] = T(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (p + u < pe_) {  // uniform
            const int j = p + u - pb;
            const T* rec = eb + (uint32_t)eid[u] * (uint32_t)EW;
            const T fe = rec[BP];
            const T ps = filt(ws, bs, rec, fe), pe = filt(we, be, rec, fe), pm = filt(wm, bm, rec, fe);
            T y[5];
            lane_y<T>(rec + BP, cm.l, y, y00);
            T dgs = T(0), dge = T(0);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
              dgs += xh[m] * gx[u][m];
              dge += y[m] * gx[u][m];
            }
            const T dg_m = cm.has_s ? dgm[u] : T(0);
            acc_hs += ps * dgs;
            acc_he += pe * dge;
            acc_hm += pm * dg_m;
            const T gate = hs * ps;
#pragma unroll
            for (int m = 0; m < 5; ++m) acc_xh[m] += gate * gx[u][m];
            T* qr = q_out + (int64_t)eid[u] * H;
            if (a.q_accum) {
              if (cm.has_u) {
                qr[cm.tu] += hs * dgs;
                qr[C + cm.tu] += he * dge;
              }
              if (cm.has_s) qr[2 * C + cm.ts] += hm * dg_m;
            } else {
              if (cm.has_u) {
                qr[cm.tu] = hs * dgs;
                qr[C + cm.tu] = he * dge;
              }
              if (cm.has_s) qr[2 * C + cm.ts] = hm * dg_m;
            }
            if (a.want_gy) {  // uniform
              const T gy = he * pe;
              T r1[3] = {T(0), T(0), T(0)}, r2[5] = {T(0), T(0), T(0), T(0), T(0)};
              if (wave_has1) {
#pragma unroll
                for (int m = 0; m < 3; ++m) r1[m] = wave_total(is1 ? gy * gx[u][m] : T(0));
              }
              if (wave_has2) {
#pragma unroll
                for (int m = 0; m < 5; ++m) r2[m] = wave_total(is2 ? gy * gx[u][m] : T(0));
              }
              if (lane == 0) {
#pragma unroll
                for (int m = 0; m < 3; ++m) red[j][wave][m] = r1[m];
#pragma unroll
                for (int m = 0; m < 5; ++m) red[j][wave][3 + m] = r2[m];
                if (wave == 0) red_eid[j] = eid[u];
              }
            }
          }
        }
      }
      if (!a.want_gy) continue;   // uniform: no harmonics' gradient wanted, nothing was put in LDS
      __syncthreads();
      for (int i = t; i < cnt * 8; i += 256) {  // a lane per (edge, harmonic): the four waves' sums in wave order
        const int j = i >> 3, m = i & 7;
        gy_out[(int64_t)red_eid[j] * 8 + m] = ((red[j][0][m] + red[j][1][m]) + red[j][2][m]) + red[j][3][m];
      }
      __syncthreads();
    }
    if (cm.has_u) {
      grad_h[n * H + cm.tu] = acc_hs;
      grad_h[n * H + C + cm.tu] = acc_he;
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < cm.nm) grad_xhat[xa.off + n * xa.node + m * xa.comp] = acc_xh[m];
    }
    if (cm.has_s) grad_h[n * H + 2 * C + cm.ts] = acc_hm;
  }
}

// dL/d[W | b] of the training-pass message: out[c, k] = sum_e q[e, c] rec[e, k], k <= roundup(B, 4) (the record head and the envelope
// column behind it), a [H x E] x [E x K] product with K <= 33 -- the library's kernels for this shape run at a third of the read rate.
// A workgroup owns TNQ_EDGES consecutive edges, thread t the channels t, t + 256, t + 512; the record columns are workgroup-uniform
// (staged in LDS TNQ_STAGE rows at a time, read back as broadcasts), q rows are read once, coalesced, four rows in flight.
// parts[chunk][H][K], summed by the caller in chunk order.
constexpr int TNQ_EDGES = 512;   // (256: the same kernel time, twice the partial blocks for the caller to add)
constexpr int TNQ_STAGE = 128;   // record rows staged in LDS at a time (scalar loads of eight rows ahead spilled 396 SGPRs: 740 us)
template <typename T, int MAXK>
__global__ void __launch_bounds__(256) k_q_wgrad(const T* __restrict__ q, const T* __restrict__ rec, int64_t E, int H, int EW, int K,
                                                 T* __restrict__ parts) {
  constexpr int KP = (MAXK + 3) & ~3;
  __shared__ T srec[TNQ_STAGE][KP];
  const int t = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * TNQ_EDGES, e1 = min(e0 + TNQ_EDGES, E);
  T acc[3][MAXK];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[i][k] = T(0);
  const int c0 = t, c1 = t + 256, c2 = t + 512;
  constexpr int U = 4;   // q rows in flight
  for (int64_t sb = e0; sb < e1; sb += TNQ_STAGE) {
    const int cnt = (int)min((int64_t)TNQ_STAGE, e1 - sb);
    __syncthreads();
    for (int i = t; i < cnt * KP; i += 256) {
      const int e = i / KP, k = i - e * KP;
      srec[e][k] = k < K ? rec[(sb + e) * EW + k] : T(0);
    }
    __syncthreads();
    for (int eb = 0; eb < cnt; eb += U) {
      T qv[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = eb + u < cnt;
        const T* qr = q + (sb + min(eb + u, cnt - 1)) * H;
        qv[u][0] = (live && c0 < H) ? qr[c0] : T(0);
        qv[u][1] = (live && c1 < H) ? qr[c1] : T(0);
        qv[u][2] = (live && c2 < H) ? qr[c2] : T(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T* r = srec[min(eb + u, cnt - 1)];   // uniform: LDS broadcast reads
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
          const T rv = r[k];
          acc[0][k] += qv[u][0] * rv;
          acc[1][k] += qv[u][1] * rv;
          acc[2][k] += qv[u][2] * rv;
        }
      }
    }
  }
  T* out = parts + (int64_t)blockIdx.x * H * K;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = t + 256 * i;
    if (c < H) {
#pragma unroll
      for (int k = 0; k < MAXK; ++k)
        if (k < K) out[(int64_t)c * K + k] = acc[i][k];
    }
  }
}

// rows [*n_valid, n_rows) of a row-major buffer of 4-byte words := 0: the per-edge products of the slots behind the true edge count of a
// capacity-sized list (a training step captured as one graph), which the walk never writes and the products over ALL rows would read
__global__ void __launch_bounds__(256) k_zero_rows_from(uint32_t* __restrict__ buf, int64_t row_words, int64_t n_rows,
                                                        const int32_t* __restrict__ n_valid) {
  int64_t first = *n_valid;
  if (first < 0) first = 0;
  for (int64_t r = first + blockIdx.x; r < n_rows; r += gridDim.x) {
    uint32_t* row = buf + r * row_words;
    for (int64_t i = threadIdx.x; i < row_words; i += 256) row[i] = 0u;
  }
}

// ---- the two second-order passes that share their filter, in one walk each (training pass, ops.DiffMessageGrad.backward) -------------
// Pass A replaces h by its cotangent u_h; pass B replaces (xhat, Y) by (u_xhat, u_Y) with the constant l = 0 harmonic counted as 0 and no
// scalar-message term.  Both use the TRUE record head, i.e. the same filter values and the same gathered rows of dL/dout: a "pair"
// kernel evaluates the filter once per edge and forms both passes' terms (h2 = u_h, xhat2 = u_xhat, eb2 = records whose harmonics are
// u_Y; only their tail is read).  Forward pair: s_out = s_in + A's scalar aggregate, x_out = x_in + A's + B's equivariant aggregates.
template <typename T, int MAXB, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_message_fwd_sb2(
    SbArgs a, const T* __restrict__ eb, const T* __restrict__ eb2, const T* __restrict__ h, const T* __restrict__ h2,
    const T* __restrict__ xhat, const T* __restrict__ xhat2, const T* __restrict__ s_in, const T* __restrict__ x_in,
    const T* __restrict__ w_rbf, const T* __restrict__ b_rbf, T* __restrict__ s_out, T* __restrict__ x_out) {
  const int B = a.B, C = a.C, F = a.F, D = a.D, H = a.H;
  const int BP = eb_bp(B), EW = BP + 12;
  const ChanMap cm = chan_map(a);
  const XAddr xa = xaddr(a.ir, a.n_nodes, cm.tu, a.xl);
  int xcomp[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) xcomp[m] = min(m, cm.nm - 1) * xa.comp;
  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w<T, MAXB>(w_rbf, b_rbf, cm.tu, B, cm.has_u, ws, bs);
  load_w<T, MAXB>(w_rbf, b_rbf, C + cm.tu, B, cm.has_u, we, be);
  load_w<T, MAXB>(w_rbf, b_rbf, 2 * C + cm.ts, B, cm.has_s, wm, bm);
  const int lane = threadIdx.x & 63;
  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t c = walk.next(); c >= 0; c = walk.next()) {
    const int32_t e0 = a.rowptr[c], e1 = a.rowptr[c + 1];
    T acc_s = T(0), acc_x[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t pb = e0; pb < e1; pb += 64) {
      const int cnt = min(64, e1 - pb);
      const int32_t slot_v = pb + min(lane, cnt - 1);
      const int32_t eid_v = a.perm ? a.perm[slot_v] : slot_v;
      const int32_t nbr_v = (int32_t)a.other[eid_v];
      for (int j = 0; j < cnt; ++j) {
        const uint32_t noff = (uint32_t)__builtin_amdgcn_readlane(nbr_v, j);
        const uint32_t eoff = (uint32_t)__builtin_amdgcn_readlane(eid_v, j) * (uint32_t)EW;
        const T* hn = h + noff * (uint32_t)H;
        const T* gn = h2 + noff * (uint32_t)H;
        const T hs = hn[cm.tu], he = hn[C + cm.tu];
        const T hs2 = gn[cm.tu], he2 = gn[C + cm.tu], hm2 = gn[2 * C + cm.ts];
        const T* xn = xhat + xa.off + (int64_t)noff * xa.node;
        const T* x2 = xhat2 + xa.off + (int64_t)noff * xa.node;
        T xv[5], xw[5];
        xv[0] = xn[0];
        xw[0] = x2[0];
        if (cm.wnm >= 3) {  // wave-uniform
          xv[1] = xn[xcomp[1]];
          xv[2] = xn[xcomp[2]];
          xw[1] = x2[xcomp[1]];
          xw[2] = x2[xcomp[2]];
        } else {
          xv[1] = xv[2] = xw[1] = xw[2] = T(0);
        }
        if (cm.wnm >= 5) {
          xv[3] = xn[xcomp[3]];
          xv[4] = xn[xcomp[4]];
          xw[3] = x2[xcomp[3]];
          xw[4] = x2[xcomp[4]];
        } else {
          xv[3] = xv[4] = xw[3] = xw[4] = T(0);
        }
        const T* rec = eb + eoff;
        const T* rec2 = eb2 + eoff;
        const T fe = rec[BP];
        const T ps = filt(ws, bs, rec, fe), pe = filt(we, be, rec, fe), pm = filt(wm, bm, rec, fe);
        acc_s += hm2 * pm;
        const T gsa = hs2 * ps, gea = he2 * pe, gsb = hs * ps, geb = he * pe;
        T y[5], y2[5];
        lane_y<T>(rec + BP, cm.l, y, T(1));
        lane_y<T>(rec2 + BP, cm.l, y2, T(0));
#pragma unroll
        for (int m = 0; m < 5; ++m) acc_x[m] += xv[m] * gsa + y[m] * gea + xw[m] * gsb + y2[m] * geb;
      }
    }
    if (cm.has_s) s_out[c * F + cm.ts] = (s_in ? s_in[c * F + cm.ts] : T(0)) + acc_s;
    if (cm.has_u) {
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < cm.nm) x_out[c * D + cm.off + m] = (x_in ? x_in[c * D + cm.off + m] : T(0)) + acc_x[m];
    }
  }
}

// Reverse pair: grad_h = pass B's dL/dh, grad_xhat = pass A's dL/dxhat, q = q_A + q_B, gy = pass A's dL/dY.
template <typename T, int MAXB, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) k_message_bwd_sbq2(
    SbArgs a, const T* __restrict__ eb, const T* __restrict__ eb2, const T* __restrict__ h, const T* __restrict__ h2,
    const T* __restrict__ xhat, const T* __restrict__ xhat2, const T* __restrict__ grad_s, const T* __restrict__ grad_x,
    const T* __restrict__ w_rbf, const T* __restrict__ b_rbf, T* __restrict__ grad_h, T* __restrict__ grad_xhat,
    T* __restrict__ q_out, T* __restrict__ gy_out) {
  __shared__ T red[SB_RED][4][8];
  __shared__ int32_t red_eid[SB_RED];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.B, C = a.C, F = a.F, D = a.D, H = a.H;
  const int BP = eb_bp(B), EW = BP + 12;
  const ChanMap cm = chan_map(a);
  const bool is1 = cm.has_u && cm.l == 1, is2 = cm.has_u && cm.l == 2;
  const bool wave_has1 = __ballot(is1) != 0ull, wave_has2 = __ballot(is2) != 0ull;
  const XAddr xa = xaddr(a.ir, a.n_nodes, cm.tu, a.xl);
  int gcomp[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) gcomp[m] = cm.off + min(m, cm.nm - 1);
  T ws[MAXB], we[MAXB], wm[MAXB], bs, be, bm;
  load_w<T, MAXB>(w_rbf, b_rbf, cm.tu, B, cm.has_u, ws, bs);
  load_w<T, MAXB>(w_rbf, b_rbf, C + cm.tu, B, cm.has_u, we, be);
  load_w<T, MAXB>(w_rbf, b_rbf, 2 * C + cm.ts, B, cm.has_s, wm, bm);
  XcdWalk walk(a.n_nodes, a.chunk);
  for (int64_t n = walk.next(); n >= 0; n = walk.next()) {
    const int32_t e0 = a.rowptr[n], e1 = a.rowptr[n + 1];
    const T hs = cm.has_u ? h[n * H + cm.tu] : T(0), he = cm.has_u ? h[n * H + C + cm.tu] : T(0);
    const T hs2 = cm.has_u ? h2[n * H + cm.tu] : T(0), he2 = cm.has_u ? h2[n * H + C + cm.tu] : T(0);
    const T hm2 = cm.has_s ? h2[n * H + 2 * C + cm.ts] : T(0);
    T xh[5], xw[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const bool own = cm.has_u && m < cm.nm;
      xh[m] = own ? xhat[xa.off + n * xa.node + m * xa.comp] : T(0);
      xw[m] = own ? xhat2[xa.off + n * xa.node + m * xa.comp] : T(0);
    }
    T acc_hs = T(0), acc_he = T(0), acc_xh[5] = {T(0), T(0), T(0), T(0), T(0)};
    for (int32_t pb = e0; pb < e1; pb += SB_RED) {
      const int32_t pe_ = min(pb + SB_RED, e1);
      const int cnt = pe_ - pb;
      const int32_t slot_v = pb + min(lane, cnt - 1);
      const int32_t eid_v = a.perm ? a.perm[slot_v] : slot_v;
      const int32_t ctr_v = (int32_t)a.other[eid_v];
      for (int32_t p = pb; p < pe_; ++p) {
        const int j = p - pb;
        const int32_t eid = __builtin_amdgcn_readlane(eid_v, j);
        const uint32_t cidx = (uint32_t)__builtin_amdgcn_readlane(ctr_v, j);
        T gx[5];
        const T dgm = grad_s[cidx * (uint32_t)F + cm.ts];
        const T* gr = grad_x + cidx * (uint32_t)D;
        gx[0] = gr[gcomp[0]];
        if (cm.wnm >= 3) {  // wave-uniform
          gx[1] = gr[gcomp[1]];
          gx[2] = gr[gcomp[2]];
        } else {
          gx[1] = gx[2] = T(0);
        }
        if (cm.wnm >= 5) {
          gx[3] = gr[gcomp[3]];
          gx[4] = gr[gcomp[4]];
        } else {
          gx[3] = gx[4] = T(0);
        }
#pragma unroll
        for (int m = 0; m < 5; ++m)
          if (m >= cm.nm) gx[m] = T(0);
        const T* rec = eb + (uint32_t)eid * (uint32_t)EW;
        const T* rec2 = eb2 + (uint32_t)eid * (uint32_t)EW;
        const T fe = rec[BP];
        const T ps = filt(ws, bs, rec, fe), pe = filt(we, be, rec, fe), pm = filt(wm, bm, rec, fe);
        T y[5], y2[5];
        lane_y<T>(rec + BP, cm.l, y, T(1));
        lane_y<T>(rec2 + BP, cm.l, y2, T(0));
        T dgs = T(0), dge = T(0), dgs2 = T(0), dge2 = T(0);
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          dgs += xh[m] * gx[m];     // pass A: the true xhat / Y against dL/dx_out
          dge += y[m] * gx[m];
          dgs2 += xw[m] * gx[m];    // pass B: their cotangents
          dge2 += y2[m] * gx[m];
        }
        const T dg_m = cm.has_s ? dgm : T(0);
        acc_hs += ps * dgs2;        // dL/dh: pass B (its scalar-message row has no term)
        acc_he += pe * dge2;
        const T gate = hs2 * ps;    // dL/dxhat: pass A
#pragma unroll
        for (int m = 0; m < 5; ++m) acc_xh[m] += gate * gx[m];
        T* qr = q_out + (int64_t)eid * H;
        if (cm.has_u) {
          qr[cm.tu] = hs2 * dgs + hs * dgs2;
          qr[C + cm.tu] = he2 * dge + he * dge2;
        }
        if (cm.has_s) qr[2 * C + cm.ts] = hm2 * dg_m;
        (void)pm;
        const T gy = he2 * pe;      // dL/dY: pass A
        T r1[3] = {T(0), T(0), T(0)}, r2[5] = {T(0), T(0), T(0), T(0), T(0)};
        if (wave_has1) {
#pragma unroll
          for (int m = 0; m < 3; ++m) r1[m] = wave_total(is1 ? gy * gx[m] : T(0));
        }
        if (wave_has2) {
#pragma unroll
          for (int m = 0; m < 5; ++m) r2[m] = wave_total(is2 ? gy * gx[m] : T(0));
        }
        if (lane == 0) {
#pragma unroll
          for (int m = 0; m < 3; ++m) red[j][wave][m] = r1[m];
#pragma unroll
          for (int m = 0; m < 5; ++m) red[j][wave][3 + m] = r2[m];
          if (wave == 0) red_eid[j] = eid;
        }
      }
      __syncthreads();
      for (int i = t; i < cnt * 8; i += 256) {
        const int j = i >> 3, m = i & 7;
        gy_out[(int64_t)red_eid[j] * 8 + m] = ((red[j][0][m] + red[j][1][m]) + red[j][2][m]) + red[j][3][m];
      }
      __syncthreads();
    }
    if (cm.has_u) {
      grad_h[n * H + cm.tu] = acc_hs;
      grad_h[n * H + C + cm.tu] = acc_he;
#pragma unroll
      for (int m = 0; m < 5; ++m)
        if (m < cm.nm) grad_xhat[xa.off + n * xa.node + m * xa.comp] = acc_xh[m];
    }
    if (cm.has_s) grad_h[n * H + 2 * C + cm.ts] = T(0);
  }
}

// what the scalar-broadcast kernels cover: at most 256 channels per kind (one thread each) and 32-bit row offsets
static bool sb_fits(int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]) {
  if (num_basis < 1 || num_basis > 32 || mul[0] < 0 || mul[1] < 0 || mul[2] < 0) return false;
  const int64_t C = (int64_t)mul[0] + mul[1] + mul[2], D = (int64_t)mul[0] + 3 * mul[1] + 5 * mul[2], H = node_dim + 2 * C;
  return C >= 1 && C <= 256 && node_dim >= 1 && node_dim <= 256 && n_nodes >= 0 && n_edges >= 0 &&
         n_nodes * (H > D ? H : D) < (1ll << 31) && n_edges * (int64_t)eb_width(num_basis) < (1ll << 31);
}

static int sb_check(const char* who, int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3],
                    SbArgs& a) {
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (1ll << 31) && n_nodes < (1ll << 31), "%s: bad sizes", who);
  XEQ_CHECK_ARG(num_basis >= 1 && num_basis <= 32, "%s: num_basis %d outside the supported range 1..32", who, num_basis);
  for (int l = 0; l < 3; ++l) {
    XEQ_CHECK_ARG(mul[l] >= 0, "%s: negative multiplicity", who);
    a.ir.mul[l] = mul[l];
  }
  a.C = a.ir.C();
  a.D = a.ir.D();
  a.F = node_dim;
  a.H = a.F + 2 * a.C;
  a.B = num_basis;
  XEQ_CHECK_ARG(a.C >= 1 && a.C <= 256 && a.F >= 1 && a.F <= 256,
                "%s: node_dim %d / %d irrep channels exceed the 256-channel workgroup mapping", who, a.F, a.C);
  XEQ_CHECK_ARG(n_nodes * (int64_t)(a.H > a.D ? a.H : a.D) < (1ll << 31) && n_edges * (int64_t)eb_width(num_basis) < (1ll << 31),
                "%s: tensors too large for 32-bit row offsets (shard the batch)", who);
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.chunk = n_nodes >= 32 * 1024 ? 32 : (int)(n_nodes / 1024 > 0 ? n_nodes / 1024 : 1);
  return XEQ_OK;
}

}  // namespace xeq

using namespace xeq;

// MAXB must cover the zero-padded record head BP = roundup(B, 4)
#define XEQ_SB_DISPATCH(KERNEL, UF, UD, ...)                                                                        \
  do {                                                                                                              \
    if (dtype == XEQ_F32) {                                                                                         \
      using T = float;                                                                                              \
      if (num_basis <= 8) hipLaunchKernelGGL((KERNEL<T, 8, UF, 4>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
      else if (num_basis <= 16) hipLaunchKernelGGL((KERNEL<T, 16, UF, 4>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else if (num_basis <= 20) hipLaunchKernelGGL((KERNEL<T, 20, UF, 4>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<T, 32, UF, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);           \
    } else if (dtype == XEQ_F64) {                                                                                  \
      using T = double;                                                                                             \
      if (num_basis <= 8) hipLaunchKernelGGL((KERNEL<T, 8, UD, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
      else if (num_basis <= 16) hipLaunchKernelGGL((KERNEL<T, 16, UD, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else if (num_basis <= 20) hipLaunchKernelGGL((KERNEL<T, 20, UD, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<T, 32, UD, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);           \
    } else {                                                                                                        \
      xeq::set_error("unsupported dtype %d", dtype);                                                                \
      return XEQ_ERR_INVALID_ARGUMENT;                                                                              \
    }                                                                                                               \
  } while (0)

extern "C" {

int xeq_edge_basis_width(int num_basis) { return eb_width(num_basis); }

int xeq_message_sb_fits(int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]) {
  return sb_fits(n_nodes, n_edges, num_basis, node_dim, mul) ? 1 : 0;
}

int xeq_edge_basis(int dtype, const void* vec, int64_t n_edges, int rbf_kind, int cutoff_kind, int num_basis,
                   double cutoff, const void* p0, const void* p1, void* basis, void* dbasis, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && num_basis >= 1 && num_basis <= 32 && cutoff > 0, "xeq_edge_basis: bad sizes");
  XEQ_CHECK_ARG(rbf_kind >= XEQ_RBF_BESSEL && rbf_kind <= XEQ_RBF_EXPNORM, "xeq_edge_basis: rbf kernel %d is not implemented", rbf_kind);
  XEQ_CHECK_ARG(rbf_kind == XEQ_RBF_BESSEL || p1 != nullptr, "xeq_edge_basis: this radial basis needs its second parameter array (std / logc / mu)");
  XEQ_CHECK_ARG(cutoff_kind == XEQ_CUTOFF_COSINE || cutoff_kind == XEQ_CUTOFF_POLYNOMIAL, "xeq_edge_basis: cutoff function %d is not implemented", cutoff_kind);
  if (n_edges == 0) return XEQ_OK;
  RadialSpec rs{rbf_kind, cutoff_kind, num_basis, cutoff};
  const int64_t total = n_edges * (num_basis + 1);
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_edge_basis<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)vec, n_edges, rs, (const T*)p0, (const T*)p1, (T*)basis, (T*)dbasis);
  });
  XEQ_CHECK_LAUNCH("xeq_edge_basis");
  return XEQ_OK;
}

int xeq_message_fwd_sb(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                       const int64_t* nbr, const void* basis, const void* h, const void* xhat, const void* s_in,
                       const void* x_in, const void* w_rbf, const void* b_rbf, int num_basis, int node_dim,
                       const int32_t mul[3], void* s_out, void* x_out, int xhat_layout, void* stream) {
  SbArgs a{};
  int rcode = sb_check("xeq_message_fwd_sb", n_nodes, n_edges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = rowptr;
  a.perm = perm;
  a.other = nbr;
  a.xl = xhat_layout & 1;   // the XEQ_XHAT_HIGHER_L_ZERO hint is for the wq kernels; this family computes the general form
  a.y0_zero = (xhat_layout & XEQ_SB_Y0_ZERO) ? 1 : 0;
  dim3 grid((unsigned)(n_nodes < 2048 ? n_nodes : 2048));
  if (dtype == XEQ_F32 && num_basis <= 20 && n_nodes <= 512 && n_nodes <= xeq_small_rows()) {
    // few nodes (an MD-sized system: at most two workgroups per CU): records staged in LDS per segment, 256 registers allowed
    using T = float;
    hipLaunchKernelGGL((k_message_fwd_sb<T, 20, 2, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, a, (const T*)basis, (const T*)h, (const T*)xhat,
                       (const T*)s_in, (const T*)x_in, (const T*)w_rbf, (const T*)b_rbf, (T*)s_out, (T*)x_out);
  } else
  XEQ_SB_DISPATCH(k_message_fwd_sb, 2, 2, a, (const T*)basis, (const T*)h, (const T*)xhat, (const T*)s_in, (const T*)x_in,
                  (const T*)w_rbf, (const T*)b_rbf, (T*)s_out, (T*)x_out);
  XEQ_CHECK_LAUNCH("xeq_message_fwd_sb");
  return XEQ_OK;
}

int xeq_message_bwd_sb(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                       const int64_t* center, const void* basis, const void* dbasis, const void* h, const void* xhat,
                       const void* grad_s, const void* grad_x, const void* w_rbf, const void* b_rbf, int num_basis,
                       int node_dim, const int32_t mul[3], void* grad_h, void* grad_xhat, void* grad_vec,
                       int xhat_layout, void* stream) {
  SbArgs a{};
  int rcode = sb_check("xeq_message_bwd_sb", n_nodes, n_edges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = n_rowptr;
  a.perm = n_perm;
  a.other = center;
  a.xl = xhat_layout & 1;   // the XEQ_XHAT_HIGHER_L_ZERO hint is for the wq kernels; this family computes the general form
  a.acc_vec = (xhat_layout & XEQ_SB_ACCUM_VEC) ? 1 : 0;
  dim3 grid((unsigned)(n_nodes < 2048 ? n_nodes : 2048));
  if (dtype == XEQ_F32 && num_basis <= 20 && n_nodes <= 512 && n_nodes <= xeq_small_rows()) {
    // few nodes (an MD-sized system: at most two workgroups per CU): the group's records staged in LDS (one round of vector loads
    // instead of a scalar-memory round trip per edge) and 256 registers allowed (no spills) -- the arithmetic is the same
    using T = float;
    hipLaunchKernelGGL((k_message_bwd_sb<T, 20, 1, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, a, (const T*)basis, (const T*)dbasis, (const T*)h,
                       (const T*)xhat, (const T*)grad_s, (const T*)grad_x, (const T*)w_rbf, (const T*)b_rbf, (T*)grad_h, (T*)grad_xhat,
                       (T*)grad_vec);
  } else
  XEQ_SB_DISPATCH(k_message_bwd_sb, 1, 1, a, (const T*)basis, (const T*)dbasis, (const T*)h, (const T*)xhat,
                  (const T*)grad_s, (const T*)grad_x, (const T*)w_rbf, (const T*)b_rbf, (T*)grad_h, (T*)grad_xhat,
                  (T*)grad_vec);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_sb");
  return XEQ_OK;
}

#define XEQ_SBQ_DISPATCH(...)                                                                                            \
  do {                                                                                                                   \
    if (dtype == XEQ_F32) {                                                                                              \
      using T = float;                                                                                                   \
      if (num_basis <= 8) hipLaunchKernelGGL((k_message_bwd_sbq<T, 8, 2, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);        \
      else if (num_basis <= 16) hipLaunchKernelGGL((k_message_bwd_sbq<T, 16, 2, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else if (num_basis <= 20) hipLaunchKernelGGL((k_message_bwd_sbq<T, 20, 2, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((k_message_bwd_sbq<T, 32, 2, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
    } else if (dtype == XEQ_F64) {                                                                                       \
      using T = double;                                                                                                  \
      if (num_basis <= 8) hipLaunchKernelGGL((k_message_bwd_sbq<T, 8, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);        \
      else if (num_basis <= 16) hipLaunchKernelGGL((k_message_bwd_sbq<T, 16, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else if (num_basis <= 20) hipLaunchKernelGGL((k_message_bwd_sbq<T, 20, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((k_message_bwd_sbq<T, 32, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
    } else {                                                                                                             \
      xeq::set_error("unsupported dtype %d", dtype);                                                                     \
      return XEQ_ERR_INVALID_ARGUMENT;                                                                                   \
    }                                                                                                                    \
  } while (0)

int xeq_message_bwd_sbq(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                        const int64_t* center, const void* basis, const void* h, const void* xhat, const void* grad_s,
                        const void* grad_x, const void* w_rbf, const void* b_rbf, int num_basis, int node_dim,
                        const int32_t mul[3], void* grad_h, void* grad_xhat, void* q, void* gy, int flags, void* stream) {
  SbArgs a{};
  int rcode = sb_check("xeq_message_bwd_sbq", n_nodes, n_edges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  XEQ_CHECK_ARG(n_edges * (int64_t)a.H < (1ll << 40), "xeq_message_bwd_sbq: q too large");
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = n_rowptr;
  a.perm = n_perm;
  a.other = center;
  a.xl = flags & 1;
  a.y0_zero = (flags & XEQ_SB_Y0_ZERO) ? 1 : 0;
  a.q_accum = (flags & XEQ_SB_Q_ACCUMULATE) ? 1 : 0;
  a.want_gy = (flags & XEQ_SB_NO_GY) ? 0 : 1;
  dim3 grid((unsigned)(n_nodes < 2048 ? n_nodes : 2048));
  XEQ_SBQ_DISPATCH(a, (const T*)basis, (const T*)h, (const T*)xhat, (const T*)grad_s, (const T*)grad_x, (const T*)w_rbf,
                   (const T*)b_rbf, (T*)grad_h, (T*)grad_xhat, (T*)q, (T*)gy);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_sbq");
  return XEQ_OK;
}

int xeq_message_q_wgrad_chunks(int64_t n_edges) { return (int)((n_edges + TNQ_EDGES - 1) / TNQ_EDGES); }

int xeq_message_q_wgrad(int dtype, const void* q, const void* basis, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3],
                        int n_chunks, void* parts, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && num_basis >= 1 && num_basis <= 32, "xeq_message_q_wgrad: bad sizes");
  const int H = node_dim + 2 * (mul[0] + mul[1] + mul[2]), BP = eb_bp(num_basis), EW = BP + 12, K = BP + 1;
  XEQ_CHECK_ARG(H >= 1 && H <= 768, "xeq_message_q_wgrad: %d filter rows exceed the 768 of the thread mapping", H);
  XEQ_CHECK_ARG(n_chunks == xeq_message_q_wgrad_chunks(n_edges), "xeq_message_q_wgrad: n_chunks %d, expected %d", n_chunks,
                xeq_message_q_wgrad_chunks(n_edges));
  if (n_edges == 0) return XEQ_OK;
  dim3 grid((unsigned)n_chunks);
  XEQ_DISPATCH_FLOAT(dtype, {
    if (K <= 9) hipLaunchKernelGGL((k_q_wgrad<T, 9>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q, (const T*)basis, n_edges, H, EW, K, (T*)parts);
    else if (K <= 17) hipLaunchKernelGGL((k_q_wgrad<T, 17>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q, (const T*)basis, n_edges, H, EW, K, (T*)parts);
    else if (K <= 21) hipLaunchKernelGGL((k_q_wgrad<T, 21>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q, (const T*)basis, n_edges, H, EW, K, (T*)parts);
    else hipLaunchKernelGGL((k_q_wgrad<T, 33>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)q, (const T*)basis, n_edges, H, EW, K, (T*)parts);
  });
  XEQ_CHECK_LAUNCH("xeq_message_q_wgrad");
  return XEQ_OK;
}

int xeq_zero_rows_from(void* buf, int64_t row_words, int64_t n_rows, const int32_t* n_valid, void* stream) {
  XEQ_CHECK_ARG(row_words >= 0 && n_rows >= 0 && n_valid != nullptr, "xeq_zero_rows_from: bad sizes");
  if (n_rows == 0 || row_words == 0) return XEQ_OK;
  hipLaunchKernelGGL(k_zero_rows_from, dim3((unsigned)(n_rows < 2048 ? n_rows : 2048)), dim3(256), 0, (hipStream_t)stream, (uint32_t*)buf,
                     row_words, n_rows, n_valid);
  XEQ_CHECK_LAUNCH("xeq_zero_rows_from");
  return XEQ_OK;
}

#define XEQ_SB2_DISPATCH(KERNEL, ...)                                                                                        \
  do {                                                                                                                       \
    if (dtype == XEQ_F32) {                                                                                                  \
      using T = float;                                                                                                       \
      if (num_basis <= 8) hipLaunchKernelGGL((KERNEL<T, 8, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);         \
      else if (num_basis <= 16) hipLaunchKernelGGL((KERNEL<T, 16, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
      else if (num_basis <= 20) hipLaunchKernelGGL((KERNEL<T, 20, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
      else hipLaunchKernelGGL((KERNEL<T, 32, 3>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                      \
    } else if (dtype == XEQ_F64) {                                                                                           \
      using T = double;                                                                                                      \
      if (num_basis <= 8) hipLaunchKernelGGL((KERNEL<T, 8, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);         \
      else if (num_basis <= 16) hipLaunchKernelGGL((KERNEL<T, 16, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
      else if (num_basis <= 20) hipLaunchKernelGGL((KERNEL<T, 20, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
      else hipLaunchKernelGGL((KERNEL<T, 32, 1>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                      \
    } else {                                                                                                                 \
      xeq::set_error("unsupported dtype %d", dtype);                                                                         \
      return XEQ_ERR_INVALID_ARGUMENT;                                                                                       \
    }                                                                                                                        \
  } while (0)

int xeq_message_fwd_sb_pair(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm, const int64_t* nbr,
                            const void* basis, const void* basis_u, const void* h, const void* u_h, const void* xhat, const void* u_xhat,
                            const void* s_in, const void* x_in, const void* w_rbf, const void* b_rbf, int num_basis, int node_dim,
                            const int32_t mul[3], void* s_out, void* x_out, int xhat_layout, void* stream) {
  SbArgs a{};
  int rcode = sb_check("xeq_message_fwd_sb_pair", n_nodes, n_edges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  XEQ_CHECK_ARG(n_nodes == 0 || (basis_u && u_h && u_xhat), "xeq_message_fwd_sb_pair: the cotangent operands are missing");
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = rowptr;
  a.perm = perm;
  a.other = nbr;
  a.xl = xhat_layout & 1;
  dim3 grid((unsigned)(n_nodes < 2048 ? n_nodes : 2048));
  XEQ_SB2_DISPATCH(k_message_fwd_sb2, a, (const T*)basis, (const T*)basis_u, (const T*)h, (const T*)u_h, (const T*)xhat, (const T*)u_xhat,
                   (const T*)s_in, (const T*)x_in, (const T*)w_rbf, (const T*)b_rbf, (T*)s_out, (T*)x_out);
  XEQ_CHECK_LAUNCH("xeq_message_fwd_sb_pair");
  return XEQ_OK;
}

int xeq_message_bwd_sbq_pair(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                             const int64_t* center, const void* basis, const void* basis_u, const void* h, const void* u_h, const void* xhat,
                             const void* u_xhat, const void* grad_s, const void* grad_x, const void* w_rbf, const void* b_rbf, int num_basis,
                             int node_dim, const int32_t mul[3], void* grad_h, void* grad_xhat, void* q, void* gy, int flags, void* stream) {
  SbArgs a{};
  int rcode = sb_check("xeq_message_bwd_sbq_pair", n_nodes, n_edges, num_basis, node_dim, mul, a);
  if (rcode != XEQ_OK) return rcode;
  XEQ_CHECK_ARG(n_nodes == 0 || (basis_u && u_h && u_xhat && q && gy), "xeq_message_bwd_sbq_pair: operands missing");
  if (n_nodes == 0) return XEQ_OK;
  a.rowptr = n_rowptr;
  a.perm = n_perm;
  a.other = center;
  a.xl = flags & 1;
  dim3 grid((unsigned)(n_nodes < 2048 ? n_nodes : 2048));
  XEQ_SB2_DISPATCH(k_message_bwd_sbq2, a, (const T*)basis, (const T*)basis_u, (const T*)h, (const T*)u_h, (const T*)xhat, (const T*)u_xhat,
                   (const T*)grad_s, (const T*)grad_x, (const T*)w_rbf, (const T*)b_rbf, (T*)grad_h, (T*)grad_xhat, (T*)q, (T*)gy);
  XEQ_CHECK_LAUNCH("xeq_message_bwd_sbq_pair");
  return XEQ_OK;
}

}  // extern "C"
