// The per-node chain between two message aggregations of XPaiNN as ONE launch per direction (round 4):
//
//   forward   XPainnUpdate.forward (nn/xpainn.py:206-231): LayerNorm + EquivariantLayerNorm (nn/o3layer.py:145-171), o3.Linear U and V
//             (:211-212), Invariant(V) (nn/o3layer.py:39-44), EquivariantDot(U, V) (:104-109), update_mlp (:215-216), dot_lin
//             (:222-223), the residual update (:225-229) -- followed, for every block but the last, by the front half of the NEXT
//             XPainnMessage.forward (nn/xpainn.py:128-139): its two norms and scalar_mlp.
//   reverse   the same chain backwards for the force evaluation (nn/basic.py:143-159): input gradients only.
//
// It replaces, per block and direction, xeq_update_uv_* + xeq_mlp2_* (x2) + xeq_linear_* + xeq_update_out_* + xeq_norm_* (six or seven
// launches that re-read [N, 480..1056] tensors between them) by one launch that reads (s, x) and writes what the message kernel and
// the reverse pass read.
//
// Design (MI355X): ACTIVATION-STATIONARY.  A wave owns 16 nodes and keeps every activation of theirs in registers in the matrix
// cores' accumulator layout (lane = node | channel quarter q, register = channel: one 32-channel tile is 8 VGPRs, register 4 g + e
// <-> channel 16 g + 4 q + e), so that the result of one product is the B operand of the next without any data movement between
// lanes (v_mfma_f32_16x16x32_bf16, D = W X with the WEIGHT tile -- 16 output rows x 32 k -- as the A operand: the k order inside a
// 32-wide step is permuted identically in the packed weights).  Every layer norm, SiLU, invariant and residual is then plain
// per-lane arithmetic (a row reduction is an in-lane sum plus two exchanges between the four lane quarters).  The weights are what
// streams: packed once per weight version in CONSUMPTION order (one linear array of 3 KB tiles, xeq_node_block_pack), pulled by the
// four waves of a workgroup through a double-buffered LDS ring (each wave fetches one tile of a four-tile stage, one barrier per
// stage) and read by every wave as A fragments.  One workgroup = 4 waves = 64 nodes; 256 registers and 72 KB of LDS, so two
// workgroups share a CU.  (The first form of this file gave a wave 32 nodes on v_mfma_f32_32x32x16_bf16, 512 registers, one wave per
// SIMD: a launch was one 510 k-cycle chain per wave whatever the node count, profiles/r04_nodeblock.txt.)
//
// Arithmetic: every contraction runs on bf16 MFMAs over operands split three ways (x = hi + mid + lo, bf16 each, exact to 24 bits),
// six products per k-step with f32 accumulation (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi; what is left out is 2^-26 of a
// product with round-to-nearest splits, below one f32 rounding): 192 instead of 512 matrix-pipe cycles per 32 x 32 x 16 tile of the
// exact-f32 MFMA.  A row's sums do not depend on the rows it shares a wave with, so sharded / chunked / padded batches agree bit for
// bit as before.
//
// Build: WITHOUT packed-fp32 instructions (csrc/build.py: -target-feature -packed-fp32-ops).  With two waves of these kernels on a SIMD,
// v_pk_fma_f32 / v_pk_add_f32 on matrix-core results gave sporadic wrong values in one 16-lane row (profiles/r04_nodeblock.txt item 9c);
// tests/test_gpu_nodeblock.py::test_node_block_at_full_size_equals_itself_on_slices is the check that shows it.
#include <stdlib.h>
#include <atomic>

#include <vector>

#include "xeq_common.h"

namespace xeq {
namespace nb {

typedef float tile_t __attribute__((ext_vector_type(8)));    // one 32-channel tile of 16 nodes: register 4 g + e <-> channel 16 g + 4 q + e
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// default XPaiNN layout (nn/model.py:57-70): node_dim 128, 128x0e + 64x1o + 32x2e
constexpr int F = 128, M0 = 128, M1 = 64, M2 = 32, C = M0 + M1 + M2, D = M0 + 3 * M1 + 5 * M2;
constexpr int HM = F + 2 * C;    // scalar_mlp output (576)
constexpr int AU = C + 2 * F;    // update_mlp output (480)
constexpr int WAVE_ROWS = 16;    // nodes of a wave
constexpr int ROWS_WG = 64;      // 4 waves x 16 nodes: the default workgroup (two per CU)
constexpr int MAX_WAVES = 8;     // waves of a workgroup: 4 .. 8 (xeq::nb::waves_for), every wave a block of 16 nodes; the first four move the weights
constexpr int TILE_U4 = 192;     // one packed weight tile (16 output rows x 32 k): 3 splits x 64 lanes x 16 B
#ifndef XEQ_NB_STAGE
#define XEQ_NB_STAGE 4
#endif
#ifndef XEQ_NB_RING
#define XEQ_NB_RING 2
#endif
constexpr int RING_STAGES = XEQ_NB_RING, STAGE_TILES = XEQ_NB_STAGE;   // tiles per stage: a multiple of 4 (each wave fetches STAGE_TILES / 4 tiles of a stage)
constexpr int PF = STAGE_TILES / 4;
constexpr int RING_BYTES = RING_STAGES * STAGE_TILES * TILE_U4 * 16;

#define NB_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The stage barrier of the weight stream.  Two ring stages: the slot written behind the barrier is the one whose last tiles this wave
// has just requested, so its LDS reads are drained first.  Three stages (-DXEQ_NB_RING=3): the slot written behind the barrier held
// the stage BEFORE the one in flight -- every wave that reaches the barrier has multiplied all of that stage's tiles (their reads were
// waited for by the products), and the stage published by the barrier was written a whole stage ago by LDS operations older than reads
// this wave has since consumed (a wave's LDS operations complete in order): no drain.
#if XEQ_NB_RING >= 3
#define NB_STAGE_BARRIER() asm volatile("s_barrier" ::: "memory")
#else
#define NB_STAGE_BARRIER() NB_LDS_BARRIER()
#endif

// Development aid (-DXEQ_NB_STAMPS): core-clock stamps at the phase boundaries of the forward / reverse kernels, per (workgroup, wave)
// (cdna_hip_programming.md section 7, in-kernel stamps).  Read back through xeq_node_block_debug_stamps.
constexpr int NB_STAMPS = 24;
__device__ unsigned long long g_nb_stamps[1024 * 4 * NB_STAMPS];
#ifdef XEQ_NB_STAMPS
#define NB_STAMP(i)                                                                                          \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                               \
    if (lane == 0 && blockIdx.x < 1024) g_nb_stamps[((int)blockIdx.x * 4 + wave) * NB_STAMPS + (i)] = t_;    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#define NB_RSTAMP(i)                                                                                         \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    unsigned long long t_;                                                                                   \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
    if (lane == 0 && blockIdx.x < 1024) g_nb_stamps[((int)blockIdx.x * 4 + wave) * NB_STAMPS + (i)] = t_;    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
#else
#define NB_STAMP(i)
#define NB_RSTAMP(i)
#endif

struct Frag {   // one k-step (32 channels) of an operand, split three ways
  bf16x8 hi, mid, lo;
};

// a 32-channel activation tile in accumulator layout as the B operand of one k-step: element j of this lane is register j
__device__ __forceinline__ Frag split_t(const tile_t& t) {
  const f32x8 v = t;
  Frag f;
  f.hi = __builtin_convertvector(v, bf16x8);
  const f32x8 r1 = v - __builtin_convertvector(f.hi, f32x8);
  f.mid = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(f.mid, f32x8);
  f.lo = __builtin_convertvector(r2, bf16x8);
  return f;
}

// acc(rows 16 S .. 16 S + 15 of the tile) += W X over one k-step: small terms first
template <int S>
__device__ __forceinline__ f32x4 half_of(const tile_t& t) { return __builtin_shufflevector(t, t, 4 * S, 4 * S + 1, 4 * S + 2, 4 * S + 3); }
template <int S>
__device__ __forceinline__ void set_half(tile_t& t, const f32x4& v) {
  t[4 * S] = v[0];
  t[4 * S + 1] = v[1];
  t[4 * S + 2] = v[2];
  t[4 * S + 3] = v[3];
}
__device__ __forceinline__ f32x4 mfma6q(f32x4 a, const Frag& w, const Frag& x) {
  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo, x.hi, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.lo, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.mid, x.mid, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.mid, x.hi, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.mid, a, 0, 0, 0);
  a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi, x.hi, a, 0, 0, 0);
  return a;
}
template <int S>
__device__ __forceinline__ void mfma6(tile_t& acc, const Frag& w, const Frag& x) {
  set_half<S>(acc, mfma6q(half_of<S>(acc), w, x));
}

// The weight stream: tiles in consumption order, four per stage, TWO stages in LDS.  A wave reads the fragments of tile t + 1 while
// it multiplies tile t (two calls of lookahead: with one or two waves on a SIMD little else hides the LDS latency).  The synchronisation
// point of stage j sits two tiles before the end of stage j - 1: there every wave has the last two tiles of stage j - 1 in registers
// and waits for its LDS reads, so after the barrier (a) stage j, written behind the previous barrier, is published, and (b) stage
// j - 1's slot is free: the stage j + 1 this wave fetched one stage ago is written into it, and the fetch of stage j + 2 is issued.
struct WStream {
  const uint4* __restrict__ g;
  uint4* ring;
  int t, n_tiles, lane, wave;
  uint4 pf[PF][3];
#ifdef XEQ_NB_STAMPS
  unsigned long long t_commit = 0, t_barrier = 0;   // cycles waiting for the fetched stage / at the stage barrier
#endif
  Frag cur, nxt;   // fragments of tiles t and t + 1
  __device__ __forceinline__ void issue(int stage) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      int tile = stage * STAGE_TILES + 4 * i + wave;
      tile = tile < n_tiles ? tile : n_tiles - 1;
      const uint4* p = g + (int64_t)tile * TILE_U4 + lane;
      pf[i][0] = p[0];
      pf[i][1] = p[64];
      pf[i][2] = p[128];
    }
  }
  __device__ __forceinline__ void commit(int stage) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      uint4* q = ring + ((stage % RING_STAGES) * STAGE_TILES + 4 * i + wave) * TILE_U4 + lane;
      q[0] = pf[i][0];
      q[64] = pf[i][1];
      q[128] = pf[i][2];
    }
  }
  __device__ __forceinline__ void read(int tile) {   // -> nxt
    const uint4* q = ring + (((tile / STAGE_TILES) % RING_STAGES) * STAGE_TILES + (tile & (STAGE_TILES - 1))) * TILE_U4 + lane;
    nxt.hi = __builtin_bit_cast(bf16x8, q[0]);
    nxt.mid = __builtin_bit_cast(bf16x8, q[64]);
    nxt.lo = __builtin_bit_cast(bf16x8, q[128]);
    __builtin_amdgcn_sched_barrier(0);   // nothing crosses: the reads are issued in front of the products of the tile before
  }
  // wave_: the wave's FETCH ROLE, 0 .. 3.  In a workgroup of more than four waves (xeq::nb::waves_for) waves 4 .. 7 take the roles of waves
  // 0 .. 3 once more: they fetch the same tile and write the same bytes into the same ring slot -- a benign duplicate (L2 / L1 hits, equal
  // values) that keeps the stream free of branches and of any address arithmetic the four-wave form does not have.  (A branch around the
  // fetch spilled 130-180 more dwords; a separate unread slot cost one more address register, and the register allocator answered by
  // spilling the IN-FLIGHT fetch registers -- a wait for the load right behind its issue: 108 -> 150 us per launch at 8 192 nodes.)
  __device__ __forceinline__ void init(const uint4* g_, uint4* ring_, int n_tiles_, int lane_, int wave_) {
    g = g_;
    ring = ring_;
    n_tiles = n_tiles_;
    lane = lane_;
    wave = wave_;
    t = 0;
    issue(0);
    commit(0);
    issue(1);
    commit(1);
    NB_LDS_BARRIER();
    issue(2);
    __builtin_amdgcn_sched_barrier(0);
    read(0);
    cur = nxt;
    read(1);
  }
  // the fragments of the next tile of the program
  __device__ __forceinline__ Frag next() {
    const Frag r = cur;
    cur = nxt;
    ++t;
#ifdef XEQ_NB_SYNC_LATE
    // development: the stage boundary ONE tile later and IN FRONT of the tile's read -- the youngest LDS read in flight is then a
    // whole tile old when the drain waits for it (two ring stages)
    if ((t & (STAGE_TILES - 1)) == STAGE_TILES - 1) {
      const int j = t / STAGE_TILES + 1;
      NB_LDS_BARRIER();
      commit(j + 1);
      issue(j + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    read(t + 1);
    return r;
#endif
    read(t + 1);
    if ((t & (STAGE_TILES - 1)) == STAGE_TILES - 2) {   // tiles t, t + 1 (the last two of their stage) are in registers or on their way
      const int j = t / STAGE_TILES + 1;
#ifdef XEQ_NB_STAMPS
      unsigned long long b0_, b1_, b2_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b0_)::"memory");
      NB_LDS_BARRIER();
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b1_)::"memory");
      commit(j + 1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b2_)::"memory");
      t_barrier += b1_ - b0_;
      t_commit += b2_ - b1_;
#else
      NB_STAGE_BARRIER();
      commit(j + 1);
#endif
      issue(j + 2);
      __builtin_amdgcn_sched_barrier(0);   // the fetch stays HERE, a whole stage ahead of its commit (the scheduler otherwise sinks it to its use)
    }
    return r;
  }
};

// acc[ot] += W[ot-th row tile, this k tile] T for NOT output tiles and one 32-channel activation tile; program order (row half, ot)
template <int NOT>
__device__ __forceinline__ void accum_tile(WStream& w, tile_t (&acc)[NOT], const tile_t& T) {
  const Frag f = split_t(T);
#pragma unroll
  for (int ot = 0; ot < NOT; ++ot) mfma6<0>(acc[ot], w.next(), f);
#pragma unroll
  for (int ot = 0; ot < NOT; ++ot) mfma6<1>(acc[ot], w.next(), f);
}

// one output tile over NK k-steps whose fragments are resident; program order (k-step, row half)
template <int NK>
__device__ __forceinline__ void out_tile(WStream& w, tile_t& acc, const Frag (&fx)[NK]) {
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    mfma6<0>(acc, w.next(), fx[k]);
    mfma6<1>(acc, w.next(), fx[k]);
  }
}
// two output tiles over NK resident k-steps; program order (k-step, row half, tile of the pair)
template <int NK>
__device__ __forceinline__ void out_pair(WStream& w, tile_t& a0, tile_t& a1, const Frag (&fx)[NK]) {
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    mfma6<0>(a0, w.next(), fx[k]);
    mfma6<0>(a1, w.next(), fx[k]);
    mfma6<1>(a0, w.next(), fx[k]);
    mfma6<1>(a1, w.next(), fx[k]);
  }
}

// A set of operand fragments (up to 4 k-steps = 128 channels of this wave's 16 nodes) parked in a wave-private LDS area: the products
// that sweep many output tiles over the same operand read it from there, step by step, instead of holding 48 registers -- the loops
// over output tiles then stay rolled and small.  (LDS operations of one wave execute in order: no barrier between put and get.)
constexpr int PARK_STEPS = 4, PARK_U4 = PARK_STEPS * TILE_U4;
constexpr int LDS_BYTES = RING_BYTES + 4 * PARK_U4 * 16;           // the default workgroup of four waves: 72 KB, two per CU
constexpr int lds_bytes(int waves) { return RING_BYTES + waves * PARK_U4 * 16; }
struct Park {
  uint4* fb;
  __device__ __forceinline__ void put(int k, const Frag& f) const {
    fb[(3 * k) * 64] = __builtin_bit_cast(uint4, f.hi);
    fb[(3 * k + 1) * 64] = __builtin_bit_cast(uint4, f.mid);
    fb[(3 * k + 2) * 64] = __builtin_bit_cast(uint4, f.lo);
  }
  __device__ __forceinline__ Frag get(int k) const {
    Frag f;
    f.hi = __builtin_bit_cast(bf16x8, fb[(3 * k) * 64]);
    f.mid = __builtin_bit_cast(bf16x8, fb[(3 * k + 1) * 64]);
    f.lo = __builtin_bit_cast(bf16x8, fb[(3 * k + 2) * 64]);
    return f;
  }
};
// one output tile over NK parked k-steps; program order (k-step, row half)
template <int NK>
__device__ __forceinline__ void out_tile_p(WStream& w, tile_t& acc, const Park& pk) {
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const Frag b = pk.get(k);   // (two waves share the SIMD: the other one covers this read; a second fragment in flight costs 12 registers)
    mfma6<0>(acc, w.next(), b);
    mfma6<1>(acc, w.next(), b);
  }
}
// two output tiles over NK parked k-steps; program order (k-step, row half, tile of the pair)
template <int NK>
__device__ __forceinline__ void out_pair_p(WStream& w, tile_t& a0, tile_t& a1, const Park& pk) {
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const Frag b = pk.get(k);
    mfma6<0>(a0, w.next(), b);
    mfma6<0>(a1, w.next(), b);
    mfma6<1>(a0, w.next(), b);
    mfma6<1>(a1, w.next(), b);
  }
}

__device__ __forceinline__ tile_t zero16() {
  tile_t z;
#pragma unroll
  for (int r = 0; r < 8; ++r) z[r] = 0.f;
  return z;
}

// 32 consecutive channels starting at c0 of a row-major row (or of a parameter vector) in accumulator layout:
// register 4 g + e <-> channel c0 + 16 g + 4 h + e (h = lane >> 4: the lane's channel quarter)
__device__ __forceinline__ tile_t ld_tile(const float* __restrict__ row, int c0, int h) {
  tile_t t;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(row + c0 + 16 * g + 4 * h);
    t[4 * g] = v.x;
    t[4 * g + 1] = v.y;
    t[4 * g + 2] = v.z;
    t[4 * g + 3] = v.w;
  }
  return t;
}
// `ok`: the lane's node exists.  Lanes beyond the last node work on the LAST node's row (row = n - 1), so their results are that row's,
// bit for bit, and -DXEQ_NB_UNCOND_STORES lets them store too (equal values to equal addresses): a store that is not behind a branch is
// a memory operation the compiler can COUNT, so the waits behind it name the loads they wait for instead of draining everything.
#ifdef XEQ_NB_UNCOND_STORES
#define NB_STORE_OK(ok) true
#else
#define NB_STORE_OK(ok) (ok)
#endif
#ifdef XEQ_NB_UNCOND_PINNED   // development: unconditional stores that the scheduler may not move anything across
#define NB_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define NB_PIN() do { } while (0)
#endif
__device__ __forceinline__ void st_tile(float* __restrict__ row, int c0, int h, const tile_t& t, bool ok) {
  if (!NB_STORE_OK(ok)) return;
  NB_PIN();
#pragma unroll
  for (int g = 0; g < 2; ++g)
    *reinterpret_cast<float4*>(row + c0 + 16 * g + 4 * h) = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
  NB_PIN();
}
// a 32-channel tile of a PARAMETER vector (norm weights, biases).  -DXEQ_NB_EXP_NOPARAM (development, timing only, wrong results): no
// memory access -- what the ~80 small parameter loads of a launch, each requested right in front of its use, cost
__device__ __forceinline__ tile_t ld_par(const float* __restrict__ vec, int c0, int h) {
#ifdef XEQ_NB_EXP_NOPARAM
  extern __shared__ uint4 lds_any_[];   // (two 16-byte LDS reads of whatever the ring holds: the cost the values would have from LDS)
  const float4 u = *reinterpret_cast<const float4*>(lds_any_ + ((c0 >> 5) & 7) * 64 + (threadIdx.x & 63));
  const float4 v = *reinterpret_cast<const float4*>(lds_any_ + (((c0 >> 5) & 7) + 8) * 64 + (threadIdx.x & 63));
  tile_t t;
  t[0] = u.x; t[1] = u.y; t[2] = u.z; t[3] = u.w; t[4] = v.x; t[5] = v.y; t[6] = v.z; t[7] = v.w;
  (void)vec; (void)h;
  return t;
#else
  return ld_tile(vec, c0, h);
#endif
}
// Tensors that only these kernels read and write (the forward launch's hand-over to the reverse launch, scratch) use a layout in which
// every wave access is 1 KB of consecutive bytes: [wave block of 16 nodes][tile][register quad][lane][4 floats] (NAT_TILE floats per
// tile and block).
constexpr int NAT_TILE = 512;
// development (timing only, wrong results): -DXEQ_NB_EXP_NOSAVE drops the forward launch's stores that only the reverse launch reads;
// -DXEQ_NB_EXP_SAVED0 lets every wave of the reverse launch read wave block 0's saved activations (cache-resident: what the
// launch would cost if recomputing them were free); -DXEQ_NB_EXP_SCRATCH0 folds the reverse launch's scratch onto 256 wave blocks
#ifdef XEQ_NB_EXP_NOSAVE
#define NB_SAVE(stmt) do { } while (0)
#else
#define NB_SAVE(stmt) do { stmt; } while (0)
#endif
__device__ __forceinline__ tile_t ld_nat(const float* __restrict__ wb, int t, int lane) {
  const float4* p = reinterpret_cast<const float4*>(wb + t * NAT_TILE) + lane;
  tile_t r;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float4 v = p[64 * g];
    r[4 * g] = v.x;
    r[4 * g + 1] = v.y;
    r[4 * g + 2] = v.z;
    r[4 * g + 3] = v.w;
  }
  return r;
}
__device__ __forceinline__ void st_nat(float* __restrict__ wb, int t, int lane, const tile_t& v) {
  float4* p = reinterpret_cast<float4*>(wb + t * NAT_TILE) + lane;
#pragma unroll
  for (int g = 0; g < 2; ++g) p[64 * g] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// tile numbers: the U|V buffer (l, component m, 0 U / 1 V, channel tile c) and equivariant rows (l, m, c)
constexpr int UV_TILES = 2 * D / 32, X_TILES = D / 32, A_TILES = AU / 32, P_TILES = C / 32, S_TILES = F / 32;
__device__ __forceinline__ constexpr int uv_tile(int l, int m, int v, int c) { return l == 0 ? 4 * v + c : (l == 1 ? 8 + 4 * m + 2 * v + c : 20 + 2 * m + v); }
__device__ __forceinline__ constexpr int x_tile(int l, int m, int c) { return l == 0 ? c : (l == 1 ? 4 + 3 * c + m : 10 + m); }

// e3nn mul_ir rows (channel-major, m-minor): the DL components of 32 channels of one l > 0 block; `blk` = the block's first float
// of the row + DL * 32 * tile
template <int DL>
__device__ __forceinline__ void ld_xm(const float* __restrict__ blk, int h, tile_t (&X)[DL]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float* p = blk + DL * (16 * g + 4 * h);
    float flat[4 * DL];
#pragma unroll
    for (int q = 0; q < DL; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
      flat[4 * q] = v.x;
      flat[4 * q + 1] = v.y;
      flat[4 * q + 2] = v.z;
      flat[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < DL; ++m) X[m][4 * g + e] = flat[DL * e + m];
  }
}
template <int DL>
__device__ __forceinline__ void st_xm(float* __restrict__ blk, int h, const tile_t (&X)[DL], bool ok) {
  if (!NB_STORE_OK(ok)) return;
  NB_PIN();
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float* p = blk + DL * (16 * g + 4 * h);
    float flat[4 * DL];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < DL; ++m) flat[DL * e + m] = X[m][4 * g + e];
#pragma unroll
    for (int q = 0; q < DL; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]);
  }
  NB_PIN();
}

__device__ __forceinline__ float sum16(const tile_t& t) { return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])); }
__device__ __forceinline__ float sumsq16(const tile_t& t, float c) {
  float a = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float d = t[r] - c;
    a = __builtin_fmaf(d, d, a);
  }
  return a;
}
// this lane's share of the sum of squares of NQ * 16 consecutive floats of a row (16-byte pieces q, q + 4, ...: the four lanes of a
// node cover the span), in a fixed order
template <int NQ>
__device__ __forceinline__ float sumsq_span(const float* __restrict__ p, int h) {
  float a = 0.f;
  const float4* p4 = reinterpret_cast<const float4*>(p) + h;
#pragma unroll 11
  for (int i = 0; i < NQ; ++i) {
    const float4 v = p4[4 * i];
    a = __builtin_fmaf(v.x, v.x, a);
    a = __builtin_fmaf(v.y, v.y, a);
    a = __builtin_fmaf(v.z, v.z, a);
    a = __builtin_fmaf(v.w, v.w, a);
  }
  return a;
}
// the node's row lives in lanes n, n + 16, n + 32, n + 48
__device__ __forceinline__ float row_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// exp: the library's expf (what the other node kernels evaluate; csrc/xeq_mlp.hip)
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {  // aten silu_backward: sig (1 + x (1 - sig))
  const float sig = 1.f / (1.f + expf(-x));
  return sig * (1.f + x * (1.f - sig));
}

// ------------------------------------------------------------------------------------------------ packed weight programs
// A program is the list of weight tiles in the order a kernel consumes them, described by segments over source matrices.
// Tile (o0 + 16 s, k0) of a source holds value(o0 + 16 s + r, k0 + 16 (j >> 2) + 4 h + (j & 3)) in element j of lane (r = lane & 15,
// h = lane >> 4): the k order of an accumulator tile used as the B operand (cdna_hip_programming.md, accumulator as the next operand).
struct Seg {
  int src;                      // source matrix
  int o0, n_ot, o_stride;       // output (row) tiles: o0 + i o_stride, in units of 32 (two 16-row MFMA tiles each: s = 0, 1)
  int k0, n_kt, k_stride;       // k tiles of 32 channels (one k-step each)
  int order;                    // 0: (kt, s, ot)   1: (ot, kt, s)
};
constexpr int MAX_SEGS = 56, MAX_SRCS = 8;
struct Src {
  const float* p;
  int ld, transposed;   // value(o, k) = transposed ? p[k ld + o] : p[o ld + k]
  float scale;
};
struct PackArgs {
  Src src[MAX_SRCS];
  Seg seg[MAX_SEGS];
  int n_segs;
  uint4* out;
};

__global__ void __launch_bounds__(64) k_nb_pack(PackArgs a) {
  int tile = blockIdx.x, si = 0;
  for (; si < a.n_segs; ++si) {
    const int nt = a.seg[si].n_ot * a.seg[si].n_kt * 2;
    if (tile < nt) break;
    tile -= nt;
  }
  if (si >= a.n_segs) {   // padding tiles behind the program: zeros
    for (int sp = 0; sp < 3; ++sp) a.out[(int64_t)blockIdx.x * TILE_U4 + sp * 64 + threadIdx.x] = make_uint4(0, 0, 0, 0);
    return;
  }
  const Seg sg = a.seg[si];
  int ot, kt, s;
  if (sg.order == 0) {
    ot = tile % sg.n_ot;
    s = (tile / sg.n_ot) & 1;
    kt = tile / (2 * sg.n_ot);
  } else {
    s = tile & 1;
    kt = (tile >> 1) % sg.n_kt;
    ot = tile / (2 * sg.n_kt);
  }
  const Src sr = a.src[sg.src];
  const int lane = threadIdx.x, r = lane & 15, h = lane >> 4;
  const int o = 32 * (sg.o0 + ot * sg.o_stride) + 16 * s + r;   // s: the row half of the 32-row output tile
  const int kb = 32 * (sg.k0 + kt * sg.k_stride);
  uint32_t hi[8], mid[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kb + 16 * (j >> 2) + 4 * h + (j & 3);
    const float v = (sr.transposed ? sr.p[(int64_t)k * sr.ld + o] : sr.p[(int64_t)o * sr.ld + k]) * sr.scale;
    const __bf16 bh = (__bf16)v;
    const float r1 = v - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    hi[j] = __builtin_bit_cast(unsigned short, bh);
    mid[j] = __builtin_bit_cast(unsigned short, bm);
    lo[j] = __builtin_bit_cast(unsigned short, bl);
  }
  uint4* o4 = a.out + (int64_t)blockIdx.x * TILE_U4 + lane;
  o4[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
  o4[64] = make_uint4(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16), mid[4] | (mid[5] << 16), mid[6] | (mid[7] << 16));
  o4[128] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

static int seg_tiles(const std::vector<Seg>& v) {
  int n = 0;
  for (const Seg& s : v) n += s.n_ot * s.n_kt * 2;
  return n;
}
static int padded_tiles(int n) { return (n + STAGE_TILES - 1) / STAGE_TILES * STAGE_TILES + STAGE_TILES; }   // + one stage of zeros: the lookahead read

// forward sources: 0 W3 = update_mlp[0].weight [F, F + C]; 1..3 [W_U | W_V] / sqrt(mul_l) as [k_in = mul_l][n_out = 2 mul_l];
// 4 dot_lin.weight [F, C]; 5 W4 = update_mlp[2].weight [AU, F]; 6 W1' = next scalar_mlp[0].weight [F, F]; 7 W2' = next scalar_mlp[2].weight [HM, F]
enum { S_W3 = 0, S_UV0 = 1, S_UV1 = 2, S_UV2 = 3, S_DOT = 4, S_W4 = 5, S_W1N = 6, S_W2N = 7 };
static void program_fwd(bool tail, std::vector<Seg>& p) {
  p.clear();
  p.push_back({S_W3, 0, 4, 1, 0, 4, 1, 0});                        // hidden += W3[:, shat]
  for (int c = 0; c < 4; ++c) {                                     // l = 0: U_c, V_c over the 4 k tiles, then v_c, p_c
    p.push_back({S_UV0, c, 2, M0 / 32, 0, 4, 1, 0});
    p.push_back({S_W3, 0, 4, 1, 4 + c, 1, 1, 0});
  }
  for (int m = 0; m < 3; ++m)                                       // l = 1: per m and channel tile
    for (int c = 0; c < 2; ++c) p.push_back({S_UV1, c, 2, M1 / 32, 0, 2, 1, 0});
  for (int c = 0; c < 2; ++c) p.push_back({S_W3, 0, 4, 1, 8 + c, 1, 1, 0});
  for (int m = 0; m < 5; ++m) p.push_back({S_UV2, 0, 2, M2 / 32, 0, 1, 1, 0});   // l = 2
  p.push_back({S_W3, 0, 4, 1, 10, 1, 1, 0});
  p.push_back({S_W4, 0, 7, 1, 0, 4, 1, 1});                        // a_vv tiles
  p.push_back({S_DOT, 0, 4, 1, 0, 7, 1, 0});                       // dot_lin over the seven p tiles
  for (int c = 0; c < 4; ++c) p.push_back({S_W4, 7 + c, 2, 4, 0, 4, 1, 0});   // (a_sv, a_ss) of scalar tile c, interleaved
  if (tail) {
    p.push_back({S_W1N, 0, 4, 1, 0, 4, 1, 0});
    for (int j = 0; j < HM / 64; ++j) p.push_back({S_W2N, 2 * j, 2, 1, 0, 4, 1, 0});   // scalar_mlp[2]: pairs of output tiles
  }
}

// ------------------------------------------------------------------------------------------------ test kernel
// y = x W^T on the primitives above (both program orders): what tests/test_gpu_nodeblock.py checks the fragment maps, the split
// arithmetic and the weight ring with.  x [n, 128], W [NOT * 32, 128]; form 0: accum_tile, form 1: out_tile.
struct LinTestArgs {
  const float* x;
  int64_t n;
  const uint4* wp;
  int n_tiles, n_ot, form;
  float* y;
};
__global__ void __launch_bounds__(256, 2) k_nb_linear_test(LinTestArgs a) {
  extern __shared__ uint4 ring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, h = lane >> 4;   // node of the wave, channel quarter
  const int64_t node = (int64_t)blockIdx.x * ROWS_WG + wave * WAVE_ROWS + n;
  const bool ok = node < a.n;
  const int64_t row = ok ? node : a.n - 1;
  NB_STAMP(0);
  NB_RSTAMP(4);
  WStream w;
  w.init(a.wp, ring, a.n_tiles, lane, wave);
  const float* xr = a.x + row * 128;
  float* yr = a.y + row * (32 * a.n_ot);
  tile_t X[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) X[t] = ld_tile(xr, 32 * t, h);
  if (a.form == 0) {   // 4 output tiles, program (kt, s, ot)
    tile_t acc[4] = {zero16(), zero16(), zero16(), zero16()};
#pragma unroll
    for (int t = 0; t < 4; ++t) accum_tile<4>(w, acc, X[t]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) st_tile(yr, 32 * ot, h, acc[ot], ok);
  } else {             // n_ot output tiles, program (ot, kt, s)
    Frag fx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fx[t] = split_t(X[t]);
    }
    NB_STAMP(1);
    if (a.form == 2) {   // pairs of output tiles, program (pair; kt, s, member)
      for (int ot = 0; ot < a.n_ot; ot += 2) {
        tile_t a0 = zero16(), a1 = zero16();
        out_pair<4>(w, a0, a1, fx);
#ifdef XEQ_NB_TEST_NOSTORE
        if (a0[0] == 12345.678f) { st_tile(yr, 32 * ot, h, a0, ok); st_tile(yr, 32 * ot + 32, h, a1, ok); }
#else
        st_tile(yr, 32 * ot, h, a0, ok);
        st_tile(yr, 32 * ot + 32, h, a1, ok);
#endif
      }
    } else
    for (int ot = 0; ot < a.n_ot; ++ot) {
      tile_t acc = zero16();
      out_tile<4>(w, acc, fx);
#ifdef XEQ_NB_TEST_NOSTORE
      if (acc[0] == 12345.678f) st_tile(yr, 32 * ot, h, acc, ok);
#else
      st_tile(yr, 32 * ot, h, acc, ok);
#endif
    }
    NB_STAMP(2);
    NB_RSTAMP(5);
  }
}

// ------------------------------------------------------------------------------------------------ forward
struct FwdArgs {
  int64_t n;
  const float *s, *x;                    // block inputs [n, F], [n, D] (e3nn mul_ir)
  const float *lnw, *lnb, *eqw, *eqb;    // norms of the update block
  const float* b_uv;                     // [2 F]: update_U.bias | update_V.bias, or NULL
  const float *b3, *b4;                  // update_mlp biases
  float eps;                             // Invariant eps
  const uint4* wp;                       // packed program
  int n_tiles;
  float* p;                              // [n, C] scratch: EquivariantDot(U, V), read back by the dot_lin sweep
  float *uv, *stats, *pre, *a, *ip;      // saved for the reverse pass: U|V pair buffer (BT), norm statistics, hidden pre-activation, update_mlp output, dot_lin output
  float *s_out, *x_out;                  // block outputs (x_out may be NULL: no consumer)
  // front half of the next message block (TAIL)
  const float *lnw2, *lnb2, *eqw2, *eqb2, *b1n, *b2n;
  float *stats2, *xhat2, *pre2, *h2;     // xhat2 in BT layout
};

template <bool TAIL>
__global__ void __launch_bounds__(64 * MAX_WAVES, 1) k_node_block_fwd(FwdArgs a) {
  extern __shared__ uint4 ring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, h = lane >> 4;   // node of the wave, channel quarter
  const int n_waves = (int)(blockDim.x >> 6);   // 4 .. 8, chosen by the host (xeq::nb::waves_for)
  const int64_t node = ((int64_t)blockIdx.x * n_waves + wave) * WAVE_ROWS + n;
  const bool ok = node < a.n;
  const int64_t row = ok ? node : a.n - 1;
  const int64_t N = a.n;
  NB_STAMP(0);
  NB_RSTAMP(17);
  WStream w;
  w.init(a.wp, ring, a.n_tiles, lane, wave & 3);
  const Park pk{ring + RING_BYTES / 16 + wave * PARK_U4 + lane};
  NB_STAMP(1);
  const float* __restrict__ srow = a.s + row * F;
  const float* __restrict__ xrow = a.x + row * D;
  const int64_t wblk = (int64_t)blockIdx.x * n_waves + wave;   // this wave's block of 16 nodes in the internal layout (= node / 16)
  float* __restrict__ pw = a.p + wblk * (P_TILES * NAT_TILE);
  float* __restrict__ uvw = a.uv + wblk * (UV_TILES * NAT_TILE);
  float* __restrict__ prew = a.pre + wblk * (S_TILES * NAT_TILE);
  float* __restrict__ ipw = a.ip + wblk * (S_TILES * NAT_TILE);
  float* __restrict__ aw = a.a + wblk * (A_TILES * NAT_TILE);
  const bool wx = a.x_out != nullptr;
  const float e1 = a.eps, e2 = a.eps * a.eps;

  tile_t HID[4] = {zero16(), zero16(), zero16(), zero16()};   // update_mlp hidden pre-activation, accumulated chunk by chunk
  float mean, rstd, mean0, rr;
  {
    tile_t X0[4];
    {  // ---- LayerNorm(s) (nn.LayerNorm: biased variance, eps 1e-5) -> first K chunk of update_mlp[0]
      tile_t S[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) S[t] = ld_tile(srow, 32 * t, h);
#pragma unroll
      for (int t = 0; t < 4; ++t) X0[t] = ld_tile(xrow, 32 * t, h);   // in flight under the products below
      mean = row_sum((sum16(S[0]) + sum16(S[1])) + (sum16(S[2]) + sum16(S[3]))) * (1.f / F);
      const float var = row_sum((sumsq16(S[0], mean) + sumsq16(S[1], mean)) + (sumsq16(S[2], mean) + sumsq16(S[3], mean))) * (1.f / F);
      rstd = 1.f / sqrtf(var + 1e-5f);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const tile_t wv = ld_par(a.lnw, 32 * t, h), bv = ld_par(a.lnb, 32 * t, h);
        tile_t sh;
#pragma unroll
        for (int r = 0; r < 8; ++r) sh[r] = (S[t][r] - mean) * rstd * wv[r] + bv[r];
        accum_tile<4>(w, HID, sh);
      }
    }
    NB_STAMP(2);
    // ---- EquivariantLayerNorm statistics (nn/o3layer.py:145-171): 0e channels centred, one rms over all channels
    mean0 = row_sum((sum16(X0[0]) + sum16(X0[1])) + (sum16(X0[2]) + sum16(X0[3]))) * (1.f / M0);
    float q = (sumsq16(X0[0], mean0) + sumsq16(X0[1], mean0)) + (sumsq16(X0[2], mean0) + sumsq16(X0[3], mean0));
    q += sumsq_span<(D - M0) / 16>(xrow + M0, h);   // the l > 0 features: plain squares, no layout needed
    rr = 1.f / sqrtf(row_sum(q) * (1.f / C) + 1e-5f);
    if (NB_STORE_OK(ok && h == 0)) *reinterpret_cast<float4*>(a.stats + 4 * row) = make_float4(mean, rstd, mean0, rr);
    NB_STAMP(3);
    // ---- l = 0: normalised features -> parked fragments
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t wv = ld_par(a.eqw, 32 * t, h), bv = ld_par(a.eqb, 32 * t, h);
      tile_t xh;
#pragma unroll
      for (int r = 0; r < 8; ++r) xh[r] = (X0[t][r] - mean0) * rr * wv[r] + bv[r];
      pk.put(t, split_t(xh));
    }
  }
  // ---- l = 0: U, V (o3.Linear with bias), v = |V|, p = U V per channel tile
  for (int c = 0; c < 4; ++c) {
    tile_t bu = zero16(), bv = zero16();
    if (a.b_uv) {
      bu = ld_par(a.b_uv, 32 * c, h);
      bv = ld_par(a.b_uv + F, 32 * c, h);
    }
    tile_t U = zero16(), V = zero16();
    out_pair_p<4>(w, U, V, pk);
    U += bu;
    V += bv;
    st_nat(uvw, uv_tile(0, 0, 0, c), lane, U);
    NB_SAVE(st_nat(uvw, uv_tile(0, 0, 1, c), lane, V));
    tile_t v, p;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      v[r] = sqrtf(__builtin_fmaf(V[r], V[r], e2)) - e1;
      p[r] = U[r] * V[r];
    }
    st_nat(pw, c, lane, p);
    accum_tile<4>(w, HID, v);
  }
  NB_STAMP(4);
  {  // ---- l = 1
    tile_t X1[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ld_xm<3>(xrow + M0 + 3 * 32 * t, h, X1[t]);
      const tile_t wv = ld_par(a.eqw, M0 + 32 * t, h);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 8; ++r) X1[t][m][r] = X1[t][m][r] * rr * wv[r];
    }
    tile_t VSQ[2] = {zero16(), zero16()}, PP[2] = {zero16(), zero16()};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      pk.put(0, split_t(X1[0][m]));
      pk.put(1, split_t(X1[1][m]));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        tile_t U = zero16(), V = zero16();
        out_pair_p<2>(w, U, V, pk);
        st_nat(uvw, uv_tile(1, m, 0, c), lane, U);
        NB_SAVE(st_nat(uvw, uv_tile(1, m, 1, c), lane, V));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          VSQ[c][r] = __builtin_fmaf(V[r], V[r], VSQ[c][r]);
          PP[c][r] = __builtin_fmaf(U[r], V[r], PP[c][r]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      st_nat(pw, 4 + c, lane, PP[c]);
      tile_t v;
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = sqrtf(VSQ[c][r] + e2) - e1;
      accum_tile<4>(w, HID, v);
    }
  }
  NB_STAMP(5);
  {  // ---- l = 2
    tile_t X2[5];
    ld_xm<5>(xrow + M0 + 3 * M1, h, X2);
    const tile_t wv = ld_par(a.eqw, M0 + M1, h);
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int r = 0; r < 8; ++r) X2[m][r] = X2[m][r] * rr * wv[r];
    tile_t VSQ = zero16(), PP = zero16();
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      pk.put(0, split_t(X2[m]));
      tile_t U = zero16(), V = zero16();
      out_pair_p<1>(w, U, V, pk);
      st_nat(uvw, uv_tile(2, m, 0, 0), lane, U);
      NB_SAVE(st_nat(uvw, uv_tile(2, m, 1, 0), lane, V));
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        VSQ[r] = __builtin_fmaf(V[r], V[r], VSQ[r]);
        PP[r] = __builtin_fmaf(U[r], V[r], PP[r]);
      }
    }
    st_nat(pw, 6, lane, PP);
    tile_t v;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = sqrtf(VSQ[r] + e2) - e1;
    accum_tile<4>(w, HID, v);
  }
  NB_STAMP(6);
  // ---- hidden layer of update_mlp: bias, SiLU -> parked fragments
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    HID[t] += ld_par(a.b3, 32 * t, h);
    NB_SAVE(st_nat(prew, t, lane, HID[t]));
    tile_t hv;
#pragma unroll
    for (int r = 0; r < 8; ++r) hv[r] = silu_f(HID[t][r]);
    pk.put(t, split_t(hv));
  }
  NB_STAMP(7);
  float* __restrict__ xor_ = wx ? a.x_out + row * D : nullptr;
  float q2 = 0.f, s0 = 0.f;   // TAIL: sum of squares of the new l > 0 features, sum of the new 0e features
  // ---- a_vv tiles and the equivariant residual update x_out = x + U a_vv (nn/xpainn.py:218-219, 229); the epilogue's operands
  // (U of this lane's own stores, x, the bias) are requested in front of the tile's products
  for (int c = 0; c < 4; ++c) {
    const tile_t b4v = ld_par(a.b4, 32 * c, h);
    tile_t U = zero16(), X0c = zero16();
    if (wx) {
      U = ld_nat(uvw, uv_tile(0, 0, 0, c), lane);
      X0c = ld_tile(xrow, 32 * c, h);
    }
    tile_t av = zero16();
    out_tile_p<4>(w, av, pk);
    av += b4v;
    NB_SAVE(st_nat(aw, c, lane, av));
    if (wx) {
      tile_t xn;
#pragma unroll
      for (int r = 0; r < 8; ++r) xn[r] = __builtin_fmaf(U[r], av[r], X0c[r]);
      st_tile(xor_, 32 * c, h, xn, ok);
      if (TAIL) s0 += sum16(xn);
    }
  }
  for (int c = 0; c < 2; ++c) {
    const tile_t b4v = ld_par(a.b4, M0 + 32 * c, h);
    tile_t X[3];
    if (wx) ld_xm<3>(xrow + M0 + 3 * 32 * c, h, X);
    tile_t av = zero16();
    out_tile_p<4>(w, av, pk);
    av += b4v;
    NB_SAVE(st_nat(aw, 4 + c, lane, av));
    if (wx) {
      tile_t U[3];   // (this lane's own stores; fetched behind the products: two waves share the SIMD, the registers are worth more than the latency)
#pragma unroll
      for (int m = 0; m < 3; ++m) U[m] = ld_nat(uvw, uv_tile(1, m, 0, c), lane);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int r = 0; r < 8; ++r) X[m][r] = __builtin_fmaf(U[m][r], av[r], X[m][r]);
        if (TAIL) q2 += sumsq16(X[m], 0.f);
      }
      st_xm<3>(xor_ + M0 + 3 * 32 * c, h, X, ok);
    }
  }
  {
    const tile_t b4v = ld_par(a.b4, M0 + M1, h);
    tile_t X[5];
    if (wx) ld_xm<5>(xrow + M0 + 3 * M1, h, X);
    tile_t av = zero16();
    out_tile_p<4>(w, av, pk);
    av += b4v;
    NB_SAVE(st_nat(aw, 6, lane, av));
    if (wx) {
      tile_t U[5];
#pragma unroll
      for (int m = 0; m < 5; ++m) U[m] = ld_nat(uvw, uv_tile(2, m, 0, 0), lane);
#pragma unroll
      for (int m = 0; m < 5; ++m) {
#pragma unroll
        for (int r = 0; r < 8; ++r) X[m][r] = __builtin_fmaf(U[m][r], av[r], X[m][r]);
        if (TAIL) q2 += sumsq16(X[m], 0.f);
      }
      st_xm<5>(xor_ + M0 + 3 * M1, h, X, ok);
    }
  }
  NB_STAMP(8);
  // ---- dot_lin over the p tiles this lane stored (nn/xpainn.py:222-223)
  tile_t IP[4] = {zero16(), zero16(), zero16(), zero16()};
  {
    tile_t pt = ld_nat(pw, 0, lane);
    for (int t = 0; t < 7; ++t) {
      const tile_t cur = pt;
      if (t + 1 < 7) pt = ld_nat(pw, t + 1, lane);
      accum_tile<4>(w, IP, cur);
    }
  }
  // ---- (a_sv, a_ss) per scalar tile and the scalar residual update s_out = s + a_sv dot_lin(p) + a_ss (nn/xpainn.py:221-228)
  tile_t SN[4];
  float* __restrict__ sor = a.s_out + row * F;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const tile_t bsv = ld_par(a.b4, C + 32 * c, h), bss = ld_par(a.b4, C + F + 32 * c, h), S = ld_tile(srow, 32 * c, h);
    NB_SAVE(st_nat(ipw, c, lane, IP[c]));
    tile_t asv = zero16(), ass = zero16();
    out_pair_p<4>(w, asv, ass, pk);
    asv += bsv;
    ass += bss;
    NB_SAVE(st_nat(aw, 7 + c, lane, asv));
    NB_SAVE(st_nat(aw, 11 + c, lane, ass));
#pragma unroll
    for (int r = 0; r < 8; ++r) SN[c][r] = (S[r] + asv[r] * IP[c][r]) + ass[r];
    st_tile(sor, 32 * c, h, SN[c], ok);
  }
  NB_STAMP(9);
  if (!TAIL) return;

  // ======== front half of the next message block (nn/xpainn.py:128-139): both norms of (s_out, x_out), scalar_mlp
  const float mean_n = row_sum((sum16(SN[0]) + sum16(SN[1])) + (sum16(SN[2]) + sum16(SN[3]))) * (1.f / F);
  const float var_n = row_sum((sumsq16(SN[0], mean_n) + sumsq16(SN[1], mean_n)) + (sumsq16(SN[2], mean_n) + sumsq16(SN[3], mean_n))) * (1.f / F);
  const float rstd_n = 1.f / sqrtf(var_n + 1e-5f);
  const float mean0_n = row_sum(s0) * (1.f / M0);
  // the new 0e features (this lane's own stores) once more for the centred second moment, the l > 0 blocks for xhat
  tile_t XN0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) XN0[t] = ld_tile(xor_, 32 * t, h);
  tile_t HN[4] = {zero16(), zero16(), zero16(), zero16()};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const tile_t wv = ld_par(a.lnw2, 32 * t, h), bv = ld_par(a.lnb2, 32 * t, h);
    tile_t sh;
#pragma unroll
    for (int r = 0; r < 8; ++r) sh[r] = (SN[t][r] - mean_n) * rstd_n * wv[r] + bv[r];
    accum_tile<4>(w, HN, sh);
  }
  NB_STAMP(10);
  const float qn = q2 + ((sumsq16(XN0[0], mean0_n) + sumsq16(XN0[1], mean0_n)) + (sumsq16(XN0[2], mean0_n) + sumsq16(XN0[3], mean0_n)));
  const float rr_n = 1.f / sqrtf(row_sum(qn) * (1.f / C) + 1e-5f);
  if (NB_STORE_OK(ok && h == 0)) *reinterpret_cast<float4*>(a.stats2 + 4 * row) = make_float4(mean_n, rstd_n, mean0_n, rr_n);
  // xhat of the next block, BT layout (block l at N base_l, row (node, m), channels contiguous)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const tile_t wv = ld_par(a.eqw2, 32 * t, h), bv = ld_par(a.eqb2, 32 * t, h);
    tile_t xh;
#pragma unroll
    for (int r = 0; r < 8; ++r) xh[r] = (XN0[t][r] - mean0_n) * rr_n * wv[r] + bv[r];
    st_tile(a.xhat2 + row * M0, 32 * t, h, xh, ok);
  }
  NB_STAMP(11);
  // hidden layer of scalar_mlp -> parked fragments
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    HN[t] += ld_par(a.b1n, 32 * t, h);
    NB_SAVE(st_nat(a.pre2 + wblk * (S_TILES * NAT_TILE), t, lane, HN[t]));
    tile_t hv;
#pragma unroll
    for (int r = 0; r < 8; ++r) hv[r] = silu_f(HN[t][r]);
    pk.put(t, split_t(hv));
  }
  NB_STAMP(12);
  // scalar_mlp[2], two output tiles at a time; the l > 0 blocks of xhat (from this lane's own x_out stores) ride along
  float* __restrict__ hrow = a.h2 + row * HM;
  {
    tile_t XC[5];
    ld_xm<5>(xor_ + M0 + 3 * M1, h, XC);
    const tile_t wv = ld_par(a.eqw2, M0 + M1, h);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
#pragma unroll
      for (int r = 0; r < 8; ++r) XC[m][r] = XC[m][r] * rr_n * wv[r];
      st_tile(a.xhat2 + N * (M0 + 3 * M1) + (row * 5 + m) * M2, 0, h, XC[m], ok);
    }
  }
  for (int j = 0; j < HM / 64; ++j) {
    const tile_t b0 = ld_par(a.b2n, 64 * j, h), b1 = ld_par(a.b2n, 64 * j + 32, h);
    tile_t XA[3];
    tile_t wa = zero16();
    if (j < 2) {
      ld_xm<3>(xor_ + M0 + 3 * 32 * j, h, XA);
      wa = ld_par(a.eqw2, M0 + 32 * j, h);
    }
    tile_t a0 = zero16(), a1 = zero16();
    out_pair_p<4>(w, a0, a1, pk);
    a0 += b0;
    a1 += b1;
    st_tile(hrow, 64 * j, h, a0, ok);
    st_tile(hrow, 64 * j + 32, h, a1, ok);
    if (j < 2) {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int r = 0; r < 8; ++r) XA[m][r] = XA[m][r] * rr_n * wa[r];
        st_tile(a.xhat2 + N * M0 + (row * 3 + m) * M1, 32 * j, h, XA[m], ok);
      }
    }
  }
  NB_STAMP(13);
  NB_RSTAMP(18);
#ifdef XEQ_NB_STAMPS
  if (lane == 0 && blockIdx.x < 1024) {
    g_nb_stamps[((int)blockIdx.x * 4 + wave) * NB_STAMPS + 14] = w.t_commit;
    g_nb_stamps[((int)blockIdx.x * 4 + wave) * NB_STAMPS + 15] = w.t_barrier;
    unsigned hw_, xcc_;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    g_nb_stamps[((int)blockIdx.x * 4 + wave) * NB_STAMPS + 16] = ((unsigned long long)xcc_ << 32) | hw_;
  }
#endif
}

// ------------------------------------------------------------------------------------------------ reverse
// backward sources: 0 W2' = next scalar_mlp[2].weight [HM, F]; 1 W1' = next scalar_mlp[0].weight [F, F]; 2 W4 = update_mlp[2].weight
// [AU, F]; 3 W3 = update_mlp[0].weight [F, F + C]; 4 dot_lin.weight [F, C] (all used transposed: out = the layer's input channel);
// 5..7 [W_U | W_V] / sqrt(mul_l) as [mul_l][2 mul_l] (rows = xhat channel = the reverse product's output)
enum { B_W2N = 0, B_W1N = 1, B_W4 = 2, B_W3 = 3, B_DOT = 4, B_UV0 = 5, B_UV1 = 6, B_UV2 = 7 };
static void program_bwd(bool tail, bool gx, std::vector<Seg>& p) {
  p.clear();
  if (tail) {
    p.push_back({B_W2N, 0, 4, 1, 0, HM / 32, 1, 0});   // g_hidden' += W2'^T[:, h tile] g_h
    p.push_back({B_W1N, 0, 4, 1, 0, 4, 1, 1});         // g_shat' tiles
  }
  if (gx) p.push_back({B_W4, 0, 4, 1, 0, 7, 1, 0});     // g_hidden += W4^T[:, a_vv tiles]
  p.push_back({B_W4, 0, 4, 1, 7, 8, 1, 0});             //           += W4^T[:, a_sv | a_ss tiles]
  p.push_back({B_W3, 0, 4, 1, 0, 4, 1, 1});             // g_shat tiles
  p.push_back({B_W3, 4, 7, 1, 0, 4, 1, 1});             // g_v tiles
  p.push_back({B_DOT, 0, 7, 1, 0, 4, 1, 1});            // g_p tiles
  for (int c = 0; c < 4; ++c) p.push_back({B_UV0, 0, 4, 1, c, 2, M0 / 32, 0});            // g_xhat_0 += W_U^T g_U_c + W_V^T g_V_c
  for (int c = 0; c < 2; ++c)
    for (int m = 0; m < 3; ++m) p.push_back({B_UV1, 0, 2, 1, c, 2, M1 / 32, 0});
  for (int m = 0; m < 5; ++m) p.push_back({B_UV2, 0, 1, 1, 0, 2, M2 / 32, 0});
}

struct BwdArgs {
  int64_t n;
  // gradients arriving: with the next block's front half (TAIL) g_h [n, HM], g_xhat (BT) and the residual-path gradients
  // g_s_in = dL/ds_out, g_x_in = dL/dx_out as the message kernel's reverse leaves them; without it g_s_in / g_x_in are the totals
  // (g_x_in NULL: zero)
  const float *g_h, *g_xhat2, *g_s_in, *g_x_in;
  const float *s_out, *x_out, *stats2, *pre2, *lnw2, *eqw2;   // TAIL: saved by the forward launch
  const float *uv, *a, *ip, *pre, *s, *x, *stats, *lnw, *eqw; // update block: saved tensors and norm weights
  float eps;
  const uint4* wp;
  int n_tiles;
  float *gxo, *gp, *gv, *gw;    // scratch: total dL/dx_out [n, D], dL/dp [n, C], dL/dv [n, C], dL/dxhat * eq_w [n, D] (e3nn layout)
  float *g_s, *g_x;             // dL/ds, dL/dx of the block's inputs
};

// LayerNorm reverse on four scalar tiles, in place: g <- rstd (dy - mean(dy) - yh mean(dy yh)) + res, dy = g w, yh = (s - mean) rstd
// (two sweeps over the row's s and weight tiles: they are re-read rather than held)
__device__ __forceinline__ void ln_bwd(tile_t (&g)[4], const float* __restrict__ srow, const float* __restrict__ lnw, float mean, float rstd,
                                       const tile_t (&res)[4], int h) {
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const tile_t wv = ld_par(lnw, 32 * t, h), sv = ld_tile(srow, 32 * t, h);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float dy = g[t][r] * wv[r];
      a1 += dy;
      a2 = __builtin_fmaf(dy, (sv[r] - mean) * rstd, a2);
    }
  }
  a1 = row_sum(a1) * (1.f / F);
  a2 = row_sum(a2) * (1.f / F);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const tile_t wv = ld_par(lnw, 32 * t, h), sv = ld_tile(srow, 32 * t, h);
#pragma unroll
    for (int r = 0; r < 8; ++r) g[t][r] = rstd * (g[t][r] * wv[r] - a1 - ((sv[r] - mean) * rstd) * a2) + res[t][r];
  }
}

template <bool TAIL, bool GX>
__global__ void __launch_bounds__(64 * MAX_WAVES, 1) k_node_block_bwd(BwdArgs a) {
  extern __shared__ uint4 ring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, h = lane >> 4;   // node of the wave, channel quarter
  const int n_waves = (int)(blockDim.x >> 6);   // 4 .. 8, chosen by the host (xeq::nb::waves_for)
  const int64_t node = ((int64_t)blockIdx.x * n_waves + wave) * WAVE_ROWS + n;
  const bool ok = node < a.n;
  const int64_t row = ok ? node : a.n - 1;
  const int64_t N = a.n;
  WStream w;
  w.init(a.wp, ring, a.n_tiles, lane, wave & 3);
  const Park pk{ring + RING_BYTES / 16 + wave * PARK_U4 + lane};
  const float e2 = a.eps * a.eps;
  const int64_t wblk = (int64_t)blockIdx.x * n_waves + wave;   // this wave's block of 16 nodes in the internal layout (= node / 16)
#ifdef XEQ_NB_EXP_SCRATCH0
  const int64_t cblk = wblk & 255;
#else
  const int64_t cblk = wblk;
#endif
  float* __restrict__ gxow = a.gxo + cblk * (X_TILES * NAT_TILE);   // total dL/dx_out (GX)
  float* __restrict__ gww = a.gw + cblk * (X_TILES * NAT_TILE);
  float* __restrict__ gpw = a.gp + cblk * (P_TILES * NAT_TILE);
  float* __restrict__ gvw = a.gv + cblk * (P_TILES * NAT_TILE);
#ifdef XEQ_NB_EXP_SAVED0
  const int64_t sblk = 0;
#else
  const int64_t sblk = wblk;
#endif
  const float* __restrict__ uvw = a.uv + sblk * (UV_TILES * NAT_TILE);
  const float* __restrict__ aw = a.a + sblk * (A_TILES * NAT_TILE);
  const float* __restrict__ prew = a.pre + sblk * (S_TILES * NAT_TILE);
  const float* __restrict__ ipw = a.ip + sblk * (S_TILES * NAT_TILE);

  tile_t GS[4];   // total dL/ds_out
  if (TAIL) {
    // ---- reverse of scalar_mlp (nn/xpainn.py:139): g_hidden' = (W2'^T g_h) silu'(pre'), g_shat' = W1'^T g_hidden'
    tile_t GH[4] = {zero16(), zero16(), zero16(), zero16()};
    const float* __restrict__ ghr = a.g_h + row * HM;
    tile_t gt = ld_tile(ghr, 0, h);
    for (int kt = 0; kt < HM / 32; ++kt) {
      const tile_t cur = gt;
      if (kt + 1 < HM / 32) gt = ld_tile(ghr, 32 * (kt + 1), h);
      accum_tile<4>(w, GH, cur);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t pv = ld_nat(a.pre2 + sblk * (S_TILES * NAT_TILE), t, lane);
      tile_t gv;
#pragma unroll
      for (int r = 0; r < 8; ++r) gv[r] = GH[t][r] * silu_grad_f(pv[r]);
      pk.put(t, split_t(gv));
    }
    tile_t gsh[4] = {zero16(), zero16(), zero16(), zero16()}, res[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      res[t] = ld_tile(a.g_s_in + row * F, 32 * t, h);
      out_tile_p<4>(w, gsh[t], pk);
    }
    const float4 st2 = *reinterpret_cast<const float4*>(a.stats2 + 4 * row);
    ln_bwd(gsh, a.s_out + row * F, a.lnw2, st2.x, st2.y, res, h);
#pragma unroll
    for (int t = 0; t < 4; ++t) GS[t] = gsh[t];
    // ---- reverse of the next block's EquivariantLayerNorm (nn/o3layer.py:145-171) on x_out: two sweeps over the row
    const float mean0 = st2.z, r2 = st2.w;
    const float* __restrict__ xo = a.x_out + row * D;
    const float* __restrict__ gxi = a.g_x_in + row * D;
    float dotp = 0.f, sgw0 = 0.f, sxc0 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t g = ld_tile(a.g_xhat2 + row * M0, 32 * t, h), wv = ld_par(a.eqw2, 32 * t, h), xv = ld_tile(xo, 32 * t, h);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float gw = g[r] * wv[r], xc = xv[r] - mean0;
        dotp = __builtin_fmaf(gw, xc, dotp);
        sgw0 += gw;
        sxc0 += xc;
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      tile_t X[3];
      ld_xm<3>(xo + M0 + 3 * 32 * t, h, X);
      const tile_t wv = ld_par(a.eqw2, M0 + 32 * t, h);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const tile_t g = ld_tile(a.g_xhat2 + N * M0 + (row * 3 + m) * M1, 32 * t, h);
#pragma unroll
        for (int r = 0; r < 8; ++r) dotp = __builtin_fmaf(g[r] * wv[r], X[m][r], dotp);
      }
    }
    {
      tile_t X[5];
      ld_xm<5>(xo + M0 + 3 * M1, h, X);
      const tile_t wv = ld_par(a.eqw2, M0 + M1, h);
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const tile_t g = ld_tile(a.g_xhat2 + N * (M0 + 3 * M1) + (row * 5 + m) * M2, 0, h);
#pragma unroll
        for (int r = 0; r < 8; ++r) dotp = __builtin_fmaf(g[r] * wv[r], X[m][r], dotp);
      }
    }
    const float coef = row_sum(dotp) * r2 * r2 * r2 * (1.f / C);
    const float gmean = (r2 * row_sum(sgw0) - coef * row_sum(sxc0)) * (1.f / M0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t g = ld_tile(a.g_xhat2 + row * M0, 32 * t, h), wv = ld_par(a.eqw2, 32 * t, h), xv = ld_tile(xo, 32 * t, h),
                   rv = ld_tile(gxi, 32 * t, h);
      tile_t o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = ((r2 * (g[r] * wv[r]) - coef * (xv[r] - mean0)) - gmean) + rv[r];
      st_nat(gxow, x_tile(0, 0, t), lane, o);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      tile_t X[3], R[3];
      ld_xm<3>(xo + M0 + 3 * 32 * t, h, X);
      ld_xm<3>(gxi + M0 + 3 * 32 * t, h, R);
      const tile_t wv = ld_par(a.eqw2, M0 + 32 * t, h);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const tile_t g = ld_tile(a.g_xhat2 + N * M0 + (row * 3 + m) * M1, 32 * t, h);
#pragma unroll
        for (int r = 0; r < 8; ++r) X[m][r] = (r2 * (g[r] * wv[r]) - coef * X[m][r]) + R[m][r];
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) st_nat(gxow, x_tile(1, m, t), lane, X[m]);
    }
    {
      tile_t X[5], R[5];
      ld_xm<5>(xo + M0 + 3 * M1, h, X);
      ld_xm<5>(gxi + M0 + 3 * M1, h, R);
      const tile_t wv = ld_par(a.eqw2, M0 + M1, h);
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const tile_t g = ld_tile(a.g_xhat2 + N * (M0 + 3 * M1) + (row * 5 + m) * M2, 0, h);
#pragma unroll
        for (int r = 0; r < 8; ++r) X[m][r] = (r2 * (g[r] * wv[r]) - coef * X[m][r]) + R[m][r];
      }
#pragma unroll
      for (int m = 0; m < 5; ++m) st_nat(gxow, x_tile(2, m, 0), lane, X[m]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) GS[t] = ld_tile(a.g_s_in + row * F, 32 * t, h);
    if (GX) {   // the totals arrive in the caller's e3nn rows: into the internal layout once
      const float* __restrict__ gxi = a.g_x_in + row * D;
#pragma unroll
      for (int t = 0; t < 4; ++t) st_nat(gxow, x_tile(0, 0, t), lane, ld_tile(gxi, 32 * t, h));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        tile_t R[3];
        ld_xm<3>(gxi + M0 + 3 * 32 * t, h, R);
#pragma unroll
        for (int m = 0; m < 3; ++m) st_nat(gxow, x_tile(1, m, t), lane, R[m]);
      }
      tile_t R[5];
      ld_xm<5>(gxi + M0 + 3 * M1, h, R);
#pragma unroll
      for (int m = 0; m < 5; ++m) st_nat(gxow, x_tile(2, m, 0), lane, R[m]);
    }
  }

  // ---- reverse of the output stage (nn/xpainn.py:218-229) into the reverse of update_mlp[2]: g_hidden += W4^T[:, chunk] g_a[chunk]
  tile_t GHID[4] = {zero16(), zero16(), zero16(), zero16()};
  if (GX) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // g_a_vv = sum_m g_x_out U
      const tile_t U = ld_nat(uvw, uv_tile(0, 0, 0, c), lane), G = ld_nat(gxow, x_tile(0, 0, c), lane);
      tile_t ga;
#pragma unroll
      for (int r = 0; r < 8; ++r) ga[r] = G[r] * U[r];
      accum_tile<4>(w, GHID, ga);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      tile_t ga = zero16();
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const tile_t U = ld_nat(uvw, uv_tile(1, m, 0, c), lane), G = ld_nat(gxow, x_tile(1, m, c), lane);
#pragma unroll
        for (int r = 0; r < 8; ++r) ga[r] = __builtin_fmaf(G[r], U[r], ga[r]);
      }
      accum_tile<4>(w, GHID, ga);
    }
    {
      tile_t ga = zero16();
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const tile_t U = ld_nat(uvw, uv_tile(2, m, 0, 0), lane), G = ld_nat(gxow, x_tile(2, m, 0), lane);
#pragma unroll
        for (int r = 0; r < 8; ++r) ga[r] = __builtin_fmaf(G[r], U[r], ga[r]);
      }
      accum_tile<4>(w, GHID, ga);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {   // g_a_sv = g_s_out ip
    const tile_t ipv = ld_nat(ipw, c, lane);
    tile_t ga;
#pragma unroll
    for (int r = 0; r < 8; ++r) ga[r] = GS[c][r] * ipv[r];
    accum_tile<4>(w, GHID, ga);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) accum_tile<4>(w, GHID, GS[c]);   // g_a_ss = g_s_out
  const float4 st = *reinterpret_cast<const float4*>(a.stats + 4 * row);
  {  // ---- g_hidden silu'(pre) -> fragments; g_shat = W3^T[:F] g_hidden -> LayerNorm reverse -> g_s; g_v = W3^T[F:] g_hidden
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t pv = ld_nat(prew, t, lane);
      tile_t gv;
#pragma unroll
      for (int r = 0; r < 8; ++r) gv[r] = GHID[t][r] * silu_grad_f(pv[r]);
      pk.put(t, split_t(gv));
    }
    {
      tile_t gsh[4] = {zero16(), zero16(), zero16(), zero16()};
#pragma unroll
      for (int t = 0; t < 4; ++t) out_tile_p<4>(w, gsh[t], pk);
      ln_bwd(gsh, a.s + row * F, a.lnw, st.x, st.y, GS, h);
#pragma unroll
      for (int t = 0; t < 4; ++t) st_tile(a.g_s + row * F, 32 * t, h, gsh[t], ok);
    }
    for (int t = 0; t < 7; ++t) {   // parked in scratch for the per-block sweeps
      tile_t acc = zero16();
      out_tile_p<4>(w, acc, pk);
      st_nat(gvw, t, lane, acc);
    }
  }
  {  // ---- g_ip = g_s_out a_sv -> fragments; g_p = dot_lin^T g_ip, seven channel tiles
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t asv = ld_nat(aw, 7 + t, lane);
      tile_t gi;
#pragma unroll
      for (int r = 0; r < 8; ++r) gi[r] = GS[t][r] * asv[r];
      pk.put(t, split_t(gi));
    }
    for (int t = 0; t < 7; ++t) {
      tile_t acc = zero16();
      out_tile_p<4>(w, acc, pk);
      st_nat(gpw, t, lane, acc);
    }
  }
  // ---- per block l: g_U = g_x_out a_vv + g_p V, g_V = g_p U + g_v V / sqrt(sum_m V^2 + eps^2) (nn/o3layer.py:39-44, 104-109),
  // g_xhat = W_U^T g_U + W_V^T g_V; times the affine weight into scratch, with the sums the norm's reverse needs
  const float mean0 = st.z, rr = st.w;
  const float* __restrict__ xrow = a.x + row * D;
  float dotp = 0.f, sgw0 = 0.f, sxc0 = 0.f;
  {
    tile_t GXH[4] = {zero16(), zero16(), zero16(), zero16()};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const tile_t U = ld_nat(uvw, uv_tile(0, 0, 0, c), lane), V = ld_nat(uvw, uv_tile(0, 0, 1, c), lane), gp = ld_nat(gpw, c, lane),
                   gv = ld_nat(gvw, c, lane);
      tile_t gU, gV;
      if (GX) {
        const tile_t G = ld_nat(gxow, x_tile(0, 0, c), lane), av = ld_nat(aw, c, lane);
#pragma unroll
        for (int r = 0; r < 8; ++r) gU[r] = __builtin_fmaf(G[r], av[r], gp[r] * V[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) gU[r] = gp[r] * V[r];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) gV[r] = __builtin_fmaf(gp[r], U[r], gv[r] / sqrtf(__builtin_fmaf(V[r], V[r], e2)) * V[r]);
      accum_tile<4>(w, GXH, gU);
      accum_tile<4>(w, GXH, gV);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const tile_t wv = ld_par(a.eqw, 32 * t, h), xv = ld_tile(xrow, 32 * t, h);
      tile_t gw;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        gw[r] = GXH[t][r] * wv[r];
        const float xc = xv[r] - mean0;
        dotp = __builtin_fmaf(gw[r], xc, dotp);
        sgw0 += gw[r];
        sxc0 += xc;
      }
      st_nat(gww, x_tile(0, 0, t), lane, gw);
    }
  }
  {
    tile_t GXH[3][2];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      GXH[m][0] = zero16();
      GXH[m][1] = zero16();
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const tile_t gp = ld_nat(gpw, 4 + c, lane), gv = ld_nat(gvw, 4 + c, lane);
      tile_t U[3], V[3], G[3];
      tile_t vv = zero16();
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        U[m] = ld_nat(uvw, uv_tile(1, m, 0, c), lane);
        V[m] = ld_nat(uvw, uv_tile(1, m, 1, c), lane);
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = __builtin_fmaf(V[m][r], V[m][r], vv[r]);
      }
      tile_t av = zero16();
      if (GX) {
#pragma unroll
        for (int m = 0; m < 3; ++m) G[m] = ld_nat(gxow, x_tile(1, m, c), lane);
        av = ld_nat(aw, 4 + c, lane);
      }
      tile_t gvn;
#pragma unroll
      for (int r = 0; r < 8; ++r) gvn[r] = gv[r] / sqrtf(vv[r] + e2);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        tile_t gU, gV;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          gU[r] = GX ? __builtin_fmaf(G[m][r], av[r], gp[r] * V[m][r]) : gp[r] * V[m][r];
          gV[r] = __builtin_fmaf(gp[r], U[m][r], gvn[r] * V[m][r]);
        }
        accum_tile<2>(w, GXH[m], gU);
        accum_tile<2>(w, GXH[m], gV);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      tile_t X[3], GW[3];
      ld_xm<3>(xrow + M0 + 3 * 32 * t, h, X);
      const tile_t wv = ld_par(a.eqw, M0 + 32 * t, h);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          GW[m][r] = GXH[m][t][r] * wv[r];
          dotp = __builtin_fmaf(GW[m][r], X[m][r], dotp);
        }
#pragma unroll
      for (int m = 0; m < 3; ++m) st_nat(gww, x_tile(1, m, t), lane, GW[m]);
    }
  }
  {
    tile_t GXH[5][1];
#pragma unroll
    for (int m = 0; m < 5; ++m) GXH[m][0] = zero16();
    const tile_t gp = ld_nat(gpw, 6, lane), gv = ld_nat(gvw, 6, lane);
    tile_t G[5];
    tile_t av = zero16();
    if (GX) {
#pragma unroll
      for (int m = 0; m < 5; ++m) G[m] = ld_nat(gxow, x_tile(2, m, 0), lane);
      av = ld_nat(aw, 6, lane);
    }
    tile_t vv = zero16();
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const tile_t V = ld_nat(uvw, uv_tile(2, m, 1, 0), lane);
#pragma unroll
      for (int r = 0; r < 8; ++r) vv[r] = __builtin_fmaf(V[r], V[r], vv[r]);
    }
    tile_t gvn;
#pragma unroll
    for (int r = 0; r < 8; ++r) gvn[r] = gv[r] / sqrtf(vv[r] + e2);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      const tile_t U = ld_nat(uvw, uv_tile(2, m, 0, 0), lane), V = ld_nat(uvw, uv_tile(2, m, 1, 0), lane);
      tile_t gU, gV;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        gU[r] = GX ? __builtin_fmaf(G[m][r], av[r], gp[r] * V[r]) : gp[r] * V[r];
        gV[r] = __builtin_fmaf(gp[r], U[r], gvn[r] * V[r]);
      }
      accum_tile<1>(w, GXH[m], gU);
      accum_tile<1>(w, GXH[m], gV);
    }
    tile_t X[5], GW[5];
    ld_xm<5>(xrow + M0 + 3 * M1, h, X);
    const tile_t wv = ld_par(a.eqw, M0 + M1, h);
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        GW[m][r] = GXH[m][0][r] * wv[r];
        dotp = __builtin_fmaf(GW[m][r], X[m][r], dotp);
      }
#pragma unroll
    for (int m = 0; m < 5; ++m) st_nat(gww, x_tile(2, m, 0), lane, GW[m]);
  }
  // ---- EquivariantLayerNorm reverse of the update block's norm: g_x = r gw - coef xc - [0e] gmean + g_x_out
  const float coef = row_sum(dotp) * rr * rr * rr * (1.f / C);
  const float gmean = (rr * row_sum(sgw0) - coef * row_sum(sxc0)) * (1.f / M0);
  float* __restrict__ gxr = a.g_x + row * D;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const tile_t gw = ld_nat(gww, x_tile(0, 0, t), lane), xv = ld_tile(xrow, 32 * t, h);
    tile_t o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (rr * gw[r] - coef * (xv[r] - mean0)) - gmean;
    if (GX) o += ld_nat(gxow, x_tile(0, 0, t), lane);
    st_tile(gxr, 32 * t, h, o, ok);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    tile_t X[3], GW[3], R[3];
    ld_xm<3>(xrow + M0 + 3 * 32 * t, h, X);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      GW[m] = ld_nat(gww, x_tile(1, m, t), lane);
      if (GX) R[m] = ld_nat(gxow, x_tile(1, m, t), lane);
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 8; ++r) X[m][r] = (rr * GW[m][r] - coef * X[m][r]) + (GX ? R[m][r] : 0.f);
    st_xm<3>(gxr + M0 + 3 * 32 * t, h, X, ok);
  }
  {
    tile_t X[5], GW[5], R[5];
    ld_xm<5>(xrow + M0 + 3 * M1, h, X);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      GW[m] = ld_nat(gww, x_tile(2, m, 0), lane);
      if (GX) R[m] = ld_nat(gxow, x_tile(2, m, 0), lane);
    }
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int r = 0; r < 8; ++r) X[m][r] = (rr * GW[m][r] - coef * X[m][r]) + (GX ? R[m][r] : 0.f);
    st_xm<5>(gxr + M0 + 3 * M1, h, X, ok);
  }
}

// more than 64 KB of dynamic LDS (weight ring + the waves' fragment areas): opt in once per process
static hipError_t raise_lds() {
  static hipError_t err = [] {
    const void* fns[] = {reinterpret_cast<const void*>(&k_node_block_fwd<true>), reinterpret_cast<const void*>(&k_node_block_fwd<false>),
                         reinterpret_cast<const void*>(&k_node_block_bwd<true, true>), reinterpret_cast<const void*>(&k_node_block_bwd<false, true>),
                         reinterpret_cast<const void*>(&k_node_block_bwd<false, false>)};
    hipError_t e = hipSuccess;
    for (const void* f : fns) {
      const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(MAX_WAVES));
      if (r != hipSuccess) e = r;
    }
    return e;
  }();
  return err;
}

// Waves per workgroup for a launch over n nodes.  A wave is one serial chain (~130 us alone on its SIMD, ~170 us when two share one),
// and a launch ends with its slowest CU: with workgroups of four waves, 18 609 nodes are 291 workgroups on 256 CUs -- 35 CUs run two
// workgroups (every SIMD of theirs two waves, 167 us) while 221 CUs are done after 131 us (scratch/bench_nodeblock_sizes.py: 16 384 nodes
// 131 us, 17 408 nodes 168 us).  Between one and two workgroups of four per CU the waves are therefore dealt out EVENLY: one workgroup
// per CU of 5 .. 8 waves (the extra waves only read the weight ring), so that no CU carries twice the load of another.  Up to one
// four-wave workgroup per CU, and from two per CU on (several rounds), workgroups of four as before.  Results do not depend on the
// choice: a wave's 16 nodes see the same weights in the same order in any workgroup.
static std::atomic<int> g_waves_override{0};   // xeq_node_block_set_waves
static int waves_for(int64_t n) {
  const char* v = getenv("XEQ_NODE_BLOCK_WAVES");   // development / tests: a fixed workgroup size (read per call: a test flips it)
  int forced = v ? atoi(v) : 0;
  if (forced >= 4 && forced <= MAX_WAVES) return forced;
  forced = g_waves_override.load(std::memory_order_relaxed);
  if (forced >= 4 && forced <= MAX_WAVES) return forced;
  const int64_t blocks16 = (n + WAVE_ROWS - 1) / WAVE_ROWS;   // waves needed
  const int64_t cus = 256;
  if (blocks16 <= 4 * cus || blocks16 > MAX_WAVES * cus) return 4;
  return (int)((blocks16 + cus - 1) / cus);
}

static bool shape_ok(int node_dim, const int32_t mul[3]) { return node_dim == F && mul[0] == M0 && mul[1] == M1 && mul[2] == M2; }

}  // namespace nb
}  // namespace xeq

using namespace xeq;
using namespace xeq::nb;

extern "C" {

/* Waves per workgroup of the launches that follow, process-wide (0: the rule of waves_for above; 4 .. 8: fixed).  -> the former value, -1
 * for a value out of range.  Results do not depend on it.  runtime.GraphedStepsInFlight asks for four: with a second step's kernels on
 * the chip the CUs that finish early are not idle, and two four-wave workgroups share a CU where one of five to eight fills it. */
int xeq_node_block_set_waves(int waves) {
  if (waves != 0 && (waves < 4 || waves > MAX_WAVES)) return -1;
  return g_waves_override.exchange(waves, std::memory_order_relaxed);
}

/* Launch policy, stated once for both fronts (nn/xpainn.py, csrc/xeq_torch.cpp): whether a force evaluation of n nodes takes the fused
 * node-block launches.  A wave owns 16 nodes and walks the whole chain of its block, so a launch takes about as long for 1 500 nodes as
 * for 9 000 (~0.1 ms: one workgroup's serial chain): below XEQ_NODE_BLOCK_MIN_NODES (default 6 144) the chain of small kernels, which
 * spreads a few tiles over many workgroups, is the faster path (whole steps, QM9-shape batches: 4 587 atoms 1.22 against 1.18 ms,
 * 9 133 atoms 1.57 against 1.69 ms; profiles/r04_nodeblock.txt).  XEQ_NODE_BLOCK=0 switches the fused launches off. */
int xeq_node_block_auto(int64_t n) {
  const char* off = getenv("XEQ_NODE_BLOCK");
  if (off && off[0] == '0') return 0;
  int64_t min_nodes = 6144;
  if (const char* v = getenv("XEQ_NODE_BLOCK_MIN_NODES")) min_nodes = atoll(v);
  return n >= min_nodes;
}

/* rows of the kernels' internal tensors (uv, p, pre, a, ip, pre_next; gxo, gp, gv, gw): whole wave blocks of 16 nodes plus one workgroup's worth
 * of slack (the last workgroup's idle waves store their padding rows) */
int64_t xeq_node_block_rows(int64_t n) { return ((n + WAVE_ROWS - 1) / WAVE_ROWS + MAX_WAVES) * WAVE_ROWS; }

int xeq_node_block_supported(int dtype, int node_dim, const int32_t mul[3]) { return dtype == XEQ_F32 && mul && shape_ok(node_dim, mul); }

/* tiles (3 KB each) of the packed forward program: kind 0 with the next message block's front half, 1 without */
int64_t xeq_node_block_fwd_tiles(int with_tail) {
  std::vector<Seg> p;
  program_fwd(with_tail != 0, p);
  return padded_tiles(seg_tiles(p));
}

static int run_pack(const std::vector<Seg>& p, const Src* srcs, int n_srcs, void* out, void* stream, const char* who) {
  XEQ_CHECK_ARG((int)p.size() <= MAX_SEGS && n_srcs <= MAX_SRCS, "%s: program too long", who);
  PackArgs pa;
  for (int i = 0; i < n_srcs; ++i) pa.src[i] = srcs[i];
  for (int i = n_srcs; i < MAX_SRCS; ++i) pa.src[i] = Src{nullptr, 0, 0, 0.f};
  for (size_t i = 0; i < p.size(); ++i) pa.seg[i] = p[i];
  pa.n_segs = (int)p.size();
  pa.out = (uint4*)out;
  const int tiles = padded_tiles(seg_tiles(p));
  hipLaunchKernelGGL(k_nb_pack, dim3(tiles), dim3(64), 0, (hipStream_t)stream, pa);
  XEQ_CHECK_LAUNCH(who);
  return XEQ_OK;
}

/* Packed forward program of one update block (+ the next message block's scalar_mlp when w1n / w2n are given).
 * w3 [F, F + C], uv_l = [W_U | W_V] / sqrt(mul_l) as [mul_l, 2 mul_l] (nn/fused.py::_packed_uv), dot [F, C], w4 [C + 2 F, F],
 * w1n [F, F], w2n [F + 2 C, F]: all row-major contiguous f32.  out: xeq_node_block_fwd_tiles(..) * 3072 bytes. */
int xeq_node_block_pack_fwd(const float* w3, const float* uv0, const float* uv1, const float* uv2, const float* dot, const float* w4,
                            const float* w1n, const float* w2n, void* out, void* stream) {
  XEQ_CHECK_ARG(w3 && uv0 && uv1 && uv2 && dot && w4 && out && ((w1n == nullptr) == (w2n == nullptr)), "xeq_node_block_pack_fwd: null buffer");
  const bool tail = w1n != nullptr;
  std::vector<Seg> p;
  program_fwd(tail, p);
  const Src srcs[8] = {{w3, F + C, 0, 1.f}, {uv0, 2 * M0, 1, 1.f}, {uv1, 2 * M1, 1, 1.f}, {uv2, 2 * M2, 1, 1.f},
                       {dot, C, 0, 1.f},    {w4, F, 0, 1.f},       {w1n, F, 0, 1.f},      {w2n, F, 0, 1.f}};
  return run_pack(p, srcs, 8, out, stream, "xeq_node_block_pack_fwd");
}

/* XPainnUpdate.forward (nn/xpainn.py:206-231) and, when h_next is given, the front half of the next XPainnMessage.forward
 * (nn/xpainn.py:128-139) in one launch.  Saved for the reverse pass in the layouts the older kernels use: uv_bt (U|V pair buffer),
 * stats [n, 4], pre [n, F], a [n, C + 2 F], ip [n, F]; outputs s_out, x_out (NULL: the equivariant output has no consumer);
 * next block: stats_next [n, 4], xhat_next (BT), pre_next [n, F], h_next [n, F + 2 C]. */
int xeq_node_block_fwd(int64_t n, const float* s, const float* x, const float* ln_w, const float* ln_b, const float* eq_w, const float* eq_b,
                       const float* b_uv, const float* b3, const float* b4, double eps, const void* packed, float* p_scratch, float* uv_bt, float* stats,
                       float* pre, float* a, float* ip, float* s_out, float* x_out, const float* ln_w_next, const float* ln_b_next,
                       const float* eq_w_next, const float* eq_b_next, const float* b1_next, const float* b2_next, float* stats_next,
                       float* xhat_next, float* pre_next, float* h_next, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) / (2 * D), "xeq_node_block_fwd: n = %lld out of range", (long long)n);
  if (n == 0) return XEQ_OK;
  const bool tail = h_next != nullptr;
  XEQ_CHECK_ARG(s && x && ln_w && ln_b && eq_w && eq_b && b3 && b4 && packed && p_scratch && uv_bt && stats && pre && a && ip && s_out,
                "xeq_node_block_fwd: null buffer");
  XEQ_CHECK_ARG(!tail || (x_out && ln_w_next && ln_b_next && eq_w_next && eq_b_next && b1_next && b2_next && stats_next && xhat_next && pre_next),
                "xeq_node_block_fwd: null buffer (next block)");
  FwdArgs fa;
  fa.n = n; fa.s = s; fa.x = x; fa.lnw = ln_w; fa.lnb = ln_b; fa.eqw = eq_w; fa.eqb = eq_b; fa.b_uv = b_uv; fa.b3 = b3; fa.b4 = b4;
  fa.eps = (float)eps; fa.wp = (const uint4*)packed; fa.n_tiles = (int)xeq_node_block_fwd_tiles(tail);
  fa.p = p_scratch; fa.uv = uv_bt; fa.stats = stats; fa.pre = pre; fa.a = a; fa.ip = ip; fa.s_out = s_out; fa.x_out = x_out;
  fa.lnw2 = ln_w_next; fa.lnb2 = ln_b_next; fa.eqw2 = eq_w_next; fa.eqb2 = eq_b_next; fa.b1n = b1_next; fa.b2n = b2_next;
  fa.stats2 = stats_next; fa.xhat2 = xhat_next; fa.pre2 = pre_next; fa.h2 = h_next;
  const int nw = waves_for(n);
  const dim3 grid((unsigned)((n + nw * WAVE_ROWS - 1) / (nw * WAVE_ROWS)));
  XEQ_CHECK_ARG(raise_lds() == hipSuccess, "xeq_node_block_fwd: cannot raise the dynamic LDS limit");
  if (tail) hipLaunchKernelGGL(k_node_block_fwd<true>, grid, dim3(64 * nw), lds_bytes(nw), (hipStream_t)stream, fa);
  else hipLaunchKernelGGL(k_node_block_fwd<false>, grid, dim3(64 * nw), lds_bytes(nw), (hipStream_t)stream, fa);
  XEQ_CHECK_LAUNCH("xeq_node_block_fwd");
  return XEQ_OK;
}

/* tiles of the packed reverse program: with_tail (the next message block's front half is reversed too), with_gx (dL/dx_out is
 * not zero: every block but the last of a force evaluation) */
int64_t xeq_node_block_bwd_tiles(int with_tail, int with_gx) {
  std::vector<Seg> p;
  program_bwd(with_tail != 0, with_gx != 0 || with_tail != 0, p);
  return padded_tiles(seg_tiles(p));
}

/* Packed reverse program: the same weight tensors as xeq_node_block_pack_fwd, contracted transposed. */
int xeq_node_block_pack_bwd(const float* w3, const float* uv0, const float* uv1, const float* uv2, const float* dot, const float* w4,
                            const float* w1n, const float* w2n, int with_gx, void* out, void* stream) {
  XEQ_CHECK_ARG(w3 && uv0 && uv1 && uv2 && dot && w4 && out && ((w1n == nullptr) == (w2n == nullptr)), "xeq_node_block_pack_bwd: null buffer");
  const bool tail = w1n != nullptr;
  std::vector<Seg> p;
  program_bwd(tail, tail || with_gx != 0, p);
  const Src srcs[8] = {{w2n, F, 1, 1.f},     {w1n, F, 1, 1.f},      {w4, F, 1, 1.f},       {w3, F + C, 1, 1.f},
                       {dot, C, 1, 1.f},     {uv0, 2 * M0, 0, 1.f}, {uv1, 2 * M1, 0, 1.f}, {uv2, 2 * M2, 0, 1.f}};
  return run_pack(p, srcs, 8, out, stream, "xeq_node_block_pack_bwd");
}

/* Reverse of xeq_node_block_fwd for a force evaluation (input gradients only, nn/basic.py:143-159).
 * With the next block's front half (g_h != NULL): g_h [n, F + 2 C] and g_xhat (BT) are the gradients of h_next / xhat_next, g_s_in /
 * g_x_in the gradients that reach s_out / x_out directly (the message kernel's residual path); s_out, x_out, stats_next, pre_next as
 * the forward launch wrote them.  Without it: g_s_in = dL/ds_out, g_x_in = dL/dx_out or NULL (zero).
 * Scratch: gxo [n, D] (unused without the front half), gp [n, C], gv [n, C], gw [n, D].  Output: g_s [n, F], g_x [n, D]. */
int xeq_node_block_bwd(int64_t n, const float* g_h, const float* g_xhat_next, const float* g_s_in, const float* g_x_in, const float* s_out,
                       const float* x_out, const float* stats_next, const float* pre_next, const float* ln_w_next, const float* eq_w_next,
                       const float* uv_bt, const float* a, const float* ip, const float* pre, const float* s, const float* x,
                       const float* stats, const float* ln_w, const float* eq_w, double eps, const void* packed, float* gxo, float* gp,
                       float* gv, float* gw, float* g_s, float* g_x, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) / (2 * D), "xeq_node_block_bwd: n = %lld out of range", (long long)n);
  if (n == 0) return XEQ_OK;
  const bool tail = g_h != nullptr, gx = tail || g_x_in != nullptr;
  XEQ_CHECK_ARG(g_s_in && uv_bt && a && ip && pre && s && x && stats && ln_w && eq_w && packed && gp && gv && gw && g_s && g_x,
                "xeq_node_block_bwd: null buffer");
  XEQ_CHECK_ARG(!tail || (g_xhat_next && g_x_in && s_out && x_out && stats_next && pre_next && ln_w_next && eq_w_next),
                "xeq_node_block_bwd: null buffer (next block)");
  XEQ_CHECK_ARG(!gx || gxo, "xeq_node_block_bwd: gxo scratch missing");
  BwdArgs b;
  b.n = n; b.g_h = g_h; b.g_xhat2 = g_xhat_next; b.g_s_in = g_s_in; b.g_x_in = g_x_in; b.s_out = s_out; b.x_out = x_out;
  b.stats2 = stats_next; b.pre2 = pre_next; b.lnw2 = ln_w_next; b.eqw2 = eq_w_next; b.uv = uv_bt; b.a = a; b.ip = ip; b.pre = pre;
  b.s = s; b.x = x; b.stats = stats; b.lnw = ln_w; b.eqw = eq_w; b.eps = (float)eps; b.wp = (const uint4*)packed;
  b.n_tiles = (int)xeq_node_block_bwd_tiles(tail, gx); b.gxo = gxo; b.gp = gp; b.gv = gv; b.gw = gw; b.g_s = g_s; b.g_x = g_x;
  const int nw = waves_for(n);
  const dim3 grid((unsigned)((n + nw * WAVE_ROWS - 1) / (nw * WAVE_ROWS)));
  XEQ_CHECK_ARG(raise_lds() == hipSuccess, "xeq_node_block_bwd: cannot raise the dynamic LDS limit");
  if (tail) hipLaunchKernelGGL((k_node_block_bwd<true, true>), grid, dim3(64 * nw), lds_bytes(nw), (hipStream_t)stream, b);
  else if (gx) hipLaunchKernelGGL((k_node_block_bwd<false, true>), grid, dim3(64 * nw), lds_bytes(nw), (hipStream_t)stream, b);
  else hipLaunchKernelGGL((k_node_block_bwd<false, false>), grid, dim3(64 * nw), lds_bytes(nw), (hipStream_t)stream, b);
  XEQ_CHECK_LAUNCH("xeq_node_block_bwd");
  return XEQ_OK;
}

/* development only (XEQ_NB_STAMPS builds): out[1024 * 4 * 24] core-clock stamps per (workgroup, wave) */
int xeq_node_block_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nb_stamps), sizeof(unsigned long long) * 1024 * 4 * NB_STAMPS) == hipSuccess ? XEQ_OK : XEQ_ERR_LAUNCH;
}

/* development / test entry: y = x W^T through the kernel primitives (x [n, 128], W [32 n_ot, 128]); form 0 needs n_ot = 4 */
int xeq_node_block_linear_test(const float* x, int64_t n, const float* w, int n_ot, int form, void* packed_scratch, float* y, void* stream) {
  XEQ_CHECK_ARG(x && w && packed_scratch && y && n > 0 && n_ot > 0 && n_ot <= 18 && (form == 1 || (form == 2 && n_ot % 2 == 0) || (form == 0 && n_ot == 4)),
                "xeq_node_block_linear_test: bad arguments");
  std::vector<Seg> p;
  if (form == 2)
    for (int j = 0; j < n_ot / 2; ++j) p.push_back({0, 2 * j, 2, 1, 0, 4, 1, 0});
  else
    p.push_back({0, 0, n_ot, 1, 0, 4, 1, form});
  const Src srcs[1] = {{w, 128, 0, 1.f}};
  if (int rc = run_pack(p, srcs, 1, packed_scratch, stream, "xeq_node_block_linear_test")) return rc;
  LinTestArgs la{x, n, (const uint4*)packed_scratch, padded_tiles(seg_tiles(p)), n_ot, form, y};
  hipLaunchKernelGGL(k_nb_linear_test, dim3((unsigned)((n + ROWS_WG - 1) / ROWS_WG)), dim3(256), RING_BYTES, (hipStream_t)stream, la);
  XEQ_CHECK_LAUNCH("xeq_node_block_linear_test");
  return XEQ_OK;
}

}  // extern "C"
