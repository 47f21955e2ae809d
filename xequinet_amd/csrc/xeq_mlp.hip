// Two-layer scalar MLPs of XPainnMessage / XPainnUpdate on the matrix cores, one launch each:
//   forward   Y  = silu(X W1^T + b1) W2^T + b2          (nn/xpainn.py:103-107 scalar_mlp, :177-181 update_mlp)
//   reverse   GX = ((G W2) * silu'(pre)) W1              (input gradients only: force evaluation, nn/basic.py:143-159)
// Both are T = E(X B1 + c1), Y = T B2 + c2 with a 128-wide hidden T that never leaves the chip.
//
// Work split: a workgroup (4 waves) owns 32 consecutive rows (nodes); few tiles, or the tiles of a short last round, are shared
// by several workgroups (TileSplit, xeq_common.h).  Exact-f32 v_mfma_f32_32x32x2_f32 tiles D[column][row]: the WEIGHT fragment is
// the A operand, so a lane holds four consecutive columns of one row per register quad and every store / LDS write is 16 bytes
// wide.  Wave w computes hidden columns [32 w, 32 w + 32) in stage 1 and the output tiles w, w + 4, ... in stage 2.  The row
// operands (shared by the waves) come from LDS: the X tile is staged in 64-column chunks (double buffered), the hidden tile is
// written once by its producers.  The weight operands are private to a wave, so they go global -> register directly from a PACKED
// copy in fragment order, packed[tile][k-group][lane][4] = W[32 tile + (lane & 31)][8 group + 4 (lane >> 5) + j], one coalesced
// 16-byte load per lane for four MFMA steps (the k order inside a group of 8 is a permutation, applied to both operands); the
// bias is one more k-group, multiplied by a row of ones.  Results do not depend on the number of rows: every row sees the same
// summation order in every form.
#include <type_traits>

#include "xeq_common.h"
#include "xeq_linear_s.h"

namespace xeq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MLP_SB() __builtin_amdgcn_sched_barrier(0)
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains this wave's global stores and prefetches (vmcnt(0))
#define MLP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int MLP_ROWS = 32;
constexpr int MLP_H = 128;        // hidden width (node_dim of the default and of every shipped configuration)
constexpr int MLP_CK = 64;        // staged columns per chunk
constexpr int MLP_XLD = MLP_CK + 4;
constexpr int MLP_TLD = MLP_H + 4;

struct MlpArgs {
  const float* X;      // [n, ldx]
  int64_t ldx, n;
  int K1, N2;
  const float* W1p;    // packed [MLP_H / 32][K1 / 8 + 1][64][4] (last group: the bias)
  const float* W2p;    // packed [N2 / 32][MLP_H / 8 + 1][64][4]
  int bias1, bias2;    // whether the bias groups are non-zero
  float* pre;          // [n, MLP_H]: written (forward) / read (reverse)
  float* Y;            // [n, ldy]
  int64_t ldy;
  TileSplit ts;        // workgroup -> (row tile, part of its output tiles)
};

__global__ void k_mlp_pack(const float* __restrict__ W, const float* __restrict__ bias, int n_out, int k_in, int transposed,
                           float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the packed buffer
  const int groups = k_in / 8 + 1;                                      // + the bias group
  const int64_t total = (int64_t)(n_out / 32) * groups * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int64_t tq = idx >> 6;
  const int q = (int)(tq % groups), t = (int)(tq / groups);
  const int n = 32 * t + (lane & 31), k0 = 8 * q + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q == groups - 1) {
    if (bias && (lane >> 5) == 0) v.x = bias[n];
  } else {
    float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) pv[j] = transposed ? W[(int64_t)(k0 + j) * n_out + n] : W[(int64_t)n * k_in + k0 + j];
  }
  reinterpret_cast<float4*>(out)[idx] = v;
}

// exp: the library's expf (<= 1 ulp, what the reference's SiLU evaluates); -DXEQ_SILU_FAST: the hardware exp2 path (~2 ulp + the
// rounding of x log2 e), kept as a development switch for the accuracy / time comparison of DESIGN section 2
#ifdef XEQ_SILU_FAST
__device__ __forceinline__ float silu_exp(float x) { return __expf(x); }
#else
__device__ __forceinline__ float silu_exp(float x) { return expf(x); }
#endif
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + silu_exp(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {  // aten silu_backward: sig (1 + x (1 - sig))
  const float sig = 1.f / (1.f + silu_exp(-x));
  return sig * (1.f + x * (1.f - sig));
}

__device__ __forceinline__ float4 keep4(bool ok, const float4& v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}


// Development aid (-DXEQ_MLP_STAMPS): core-clock and 100 MHz real-time stamps around the body of every workgroup, summed;
// the quotient is the clock the chip holds under this kernel (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ unsigned long long g_mlp_stamps[8];
__device__ unsigned long long g_mlp_wg[4096 * 4];   // XEQ_MLP_STAMPS: per workgroup (hw id, xcc id, real-time start, end)
#ifdef XEQ_MLP_STAMPS
#define MLP_STAMP(core, real)                                                                \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(core), "=s"(real)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#endif

template <bool REVERSE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) k_mlp2(MlpArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[2][MLP_ROWS * MLP_XLD];
  __shared__ __attribute__((aligned(16))) float Ts[MLP_ROWS * MLP_TLD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches, scalar tile addresses
  const int i = lane & 31, kh = lane >> 5;
  int tile_, part_, parts_;
  a.ts.decode((int)blockIdx.x, tile_, part_, parts_);
  const int64_t row0 = (int64_t)tile_ * MLP_ROWS;
#ifdef XEQ_MLP_STAMPS
  unsigned long long c0_, r0_, c1_, r1_;
  MLP_STAMP(c0_, r0_);
#endif
#ifdef XEQ_MLP_WGREC   // development: start / end of every workgroup (wave 0) in 100 MHz ticks, and where it ran
  unsigned long long wg_t0_;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0_)::"memory");
#endif

  // ---- stage 1: T[32, 128] = X[32, K1] B1 + c1, hidden tile `wave` ---------------------------------------------
  const int n_chunks = (a.K1 + MLP_CK - 1) / MLP_CK;
  const int g1 = a.K1 / 8;
  // staging: 32 rows x 16 float4 per chunk = 512 float4, two per thread
  const int sr = tid >> 4, sc = (tid & 15) * 4;
  // global addresses: a workgroup-uniform 64-bit base plus a 32-bit lane offset (rows of one tile span < 2^31 elements);
  // every load is issued unconditionally from a clamped address (no branches around loads: the waits then count loads in
  // issue order and the prefetches stay in flight), out-of-range rows / columns are zeroed on their way into LDS
  const int rows_here = (int)min((int64_t)MLP_ROWS, a.n - row0);
  const float* __restrict__ xb = a.X + row0 * a.ldx;
  float* __restrict__ preb = a.pre + row0 * MLP_H;
  float* __restrict__ yo = a.Y + row0 * a.ldy;
  const unsigned ldx32 = (unsigned)a.ldx, ldy32 = (unsigned)a.ldy;
  const unsigned r0c = (unsigned)min(sr, rows_here - 1) * ldx32, r1c = (unsigned)min(sr + 16, rows_here - 1) * ldx32;
  const bool r0ok = sr < rows_here, r1ok = sr + 16 < rows_here;
  const bool row_ok = i < rows_here;
  const unsigned ic = (unsigned)min(i, rows_here - 1);
  auto fetch = [&](int c, float4& v0, float4& v1) {
    const int colc = min(MLP_CK * c + sc, a.K1 - 4);
    v0 = *reinterpret_cast<const float4*>(xb + (r0c + (unsigned)colc));
    v1 = *reinterpret_cast<const float4*>(xb + (r1c + (unsigned)colc));
  };
  auto stash = [&](int c, const float4& v0, const float4& v1) {
    const bool cok = MLP_CK * c + sc < a.K1;
    *reinterpret_cast<float4*>(&Xs[c & 1][sr * MLP_XLD + sc]) = keep4(r0ok && cok, v0);
    *reinterpret_cast<float4*>(&Xs[c & 1][(sr + 16) * MLP_XLD + sc]) = keep4(r1ok && cok, v1);
  };
  const float4* w1 = reinterpret_cast<const float4*>(a.W1p) + (int64_t)wave * (g1 + 1) * 64;   // wave-uniform base, lane offset in the load
  auto fetch_w = [&](float4 (&b)[8], int g0) {   // four k-groups from g0 on
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = w1[min(g0 + q, g1 - 1) * 64 + lane];   // past K1: any finite weight, the staged rows are zero there
  };

  // accumulator layout (weights as the A operand): lane (node = i, kh), register 4 g + e <-> column 8 g + 4 kh + e of the
  // tile: four consecutive columns of one row per register quad, i.e. 16-byte stores / LDS writes straight from registers
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.f;
    acc1[r] = 0.f;
  }
  // the bias rides in the product: one extra k step whose weight fragment is the bias (k slot 0) against a row of ones,
  // so it arrives with the weight stream instead of as a separate load that later waits behind this wave's stores
  const float one_k0 = kh == 0 ? 1.f : 0.f;
  float bias_a = 0.f;
  if (a.bias1) bias_a = reinterpret_cast<const float*>(w1 + (int64_t)g1 * 64 + lane)[0];
  float4 pv[4];   // reverse: silu'(pre) needs the forward's pre-activations of this tile; fetched under the chunk loop
  if (REVERSE) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pv[g] = *reinterpret_cast<const float4*>(preb + (ic * MLP_H + (unsigned)(32 * wave + 8 * g + 4 * kh)));
  }
  float4 x0, x1, B0[8], B1[8];   // weight fragments: ping-pong buffers, fetched half a chunk (stage 1) / a quarter pass (stage 2) ahead
  fetch(0, x0, x1);
  fetch_w(B0, 0);
  stash(0, x0, x1);
  MLP_LDS_BARRIER();
#ifdef XEQ_MLP_STAMPS
  unsigned long long cp1_, rp1_; MLP_STAMP(cp1_, rp1_);
#endif
  for (int c = 0; c < n_chunks; ++c) {
    const bool more = c + 1 < n_chunks;
    if (more) fetch(c + 1, x0, x1);
    const float* xs = &Xs[c & 1][i * MLP_XLD + 4 * kh];
    auto half = [&](const float4 (&b)[8], int h) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + 8 * (4 * h + q));   // columns past K1 are staged as zeros
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, xv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, xv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, xv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, xv.w, acc1, 0, 0, 0);
      }
    };
    fetch_w(B1, 8 * c + 4);
    MLP_SB();   // keep the fetches ahead of the MFMA block they overlap (the scheduler otherwise sinks them to their use)
    half(B0, 0);
    MLP_SB();
    fetch_w(B0, 8 * c + 8);
    MLP_SB();
    half(B1, 1);
    MLP_SB();
    if (more) stash(c + 1, x0, x1);
    MLP_LDS_BARRIER();
  }
  if (a.bias1) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, one_k0, acc0, 0, 0, 0);
#ifdef XEQ_MLP_STAMPS
  unsigned long long cp2_, rp2_; MLP_STAMP(cp2_, rp2_);
#endif

  // stage 2 walks this wave's output tiles wave, wave + 4, ... two at a time (pairs share the row fragments), each tile
  // pass in four quarters of 4 k-groups whose weights are fetched one quarter ahead (ping-pong register buffers).
  const int nt2 = a.N2 / 32;
  constexpr int G2 = MLP_H / 8;  // 16 k-groups
  const float4* w2 = reinterpret_cast<const float4*>(a.W2p);
  auto fetch_q = [&](float4 (&b)[8], int t0, bool two, int qq) {
    const float4* wa = w2 + ((int64_t)t0 * (G2 + 1) + 4 * qq) * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = wa[q * 64 + lane];
    if (two) {
      const float4* wb = wa + (int64_t)4 * (G2 + 1) * 64;
#pragma unroll
      for (int q = 0; q < 4; ++q) b[4 + q] = wb[q * 64 + lane];
    }
  };
  // split tiles (TileSplit, xeq_common.h): `parts` workgroups share the row tile, each redoes stage 1 and takes every parts-th
  // group of four output tiles (one per wave), so one tile's latency is not one workgroup's whole serial MFMA chain
  const int parts = __builtin_amdgcn_readfirstlane(parts_), part = __builtin_amdgcn_readfirstlane(part_);
  int t0 = wave + 4 * part;
  if (t0 < nt2) fetch_q(B0, t0, parts == 1 && t0 + 4 < nt2, 0);

  // elementwise stage on the hidden tile, then to LDS as the row operand of stage 2
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int col = 32 * wave + 8 * g + 4 * kh;
    float4 t = make_float4(acc0[4 * g] + acc1[4 * g], acc0[4 * g + 1] + acc1[4 * g + 1], acc0[4 * g + 2] + acc1[4 * g + 2],
                           acc0[4 * g + 3] + acc1[4 * g + 3]);
    float4 v;
    if (!REVERSE) {
      if (row_ok && part_ == 0) *reinterpret_cast<float4*>(preb + ((unsigned)i * MLP_H + (unsigned)col)) = t;
      v = make_float4(silu_f(t.x), silu_f(t.y), silu_f(t.z), silu_f(t.w));
    } else {
      v = make_float4(t.x * silu_grad_f(pv[g].x), t.y * silu_grad_f(pv[g].y), t.z * silu_grad_f(pv[g].z), t.w * silu_grad_f(pv[g].w));
    }
    *reinterpret_cast<float4*>(&Ts[i * MLP_TLD + col]) = v;
  }
  MLP_LDS_BARRIER();
#ifdef XEQ_MLP_STAMPS
  unsigned long long cp3_, cx_, cy_, rx_, st2c_ = 0; MLP_STAMP(cp3_, rx_);
#endif

  const float* ts = &Ts[i * MLP_TLD + 4 * kh];
  auto pass = [&](auto two_c, int tt, int tn, bool tn_two) {   // tn: the tile (pair) this wave takes next
    constexpr bool TWO = decltype(two_c)::value;
    f32x16 ya, yb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ya[r] = 0.f;
      yb[r] = 0.f;
    }
    float bias_ya = 0.f, bias_yb = 0.f;
    if (a.bias2) {
      bias_ya = reinterpret_cast<const float*>(w2 + ((int64_t)tt * (G2 + 1) + G2) * 64 + lane)[0];
      if (TWO) bias_yb = reinterpret_cast<const float*>(w2 + ((int64_t)(tt + 4) * (G2 + 1) + G2) * 64 + lane)[0];
    }
#ifdef XEQ_MLP_STAMPS
    MLP_STAMP(cy_, rx_);
#endif
    auto quarter = [&](const float4 (&b)[8], int qq) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 tv = *reinterpret_cast<const float4*>(ts + 8 * (4 * qq + q));
        if (TWO) {
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, tv.x, ya, 0, 0, 0);
          yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].x, tv.x, yb, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, tv.y, ya, 0, 0, 0);
          yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].y, tv.y, yb, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, tv.z, ya, 0, 0, 0);
          yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].z, tv.z, yb, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, tv.w, ya, 0, 0, 0);
          yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[4 + q].w, tv.w, yb, 0, 0, 0);
        } else {
          // one chain, the order of the paired form (a tile's sums do not depend on which form computed it): the f32 MFMA's
          // dependent-accumulator latency equals its issue interval (64 cycles)
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, tv.x, ya, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, tv.y, ya, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, tv.z, ya, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, tv.w, ya, 0, 0, 0);
        }
      }
    };
    fetch_q(B1, tt, TWO, 1);
    MLP_SB();
    quarter(B0, 0);
    MLP_SB();
    fetch_q(B0, tt, TWO, 2);
    MLP_SB();
    quarter(B1, 1);
    MLP_SB();
    fetch_q(B1, tt, TWO, 3);
    MLP_SB();
    quarter(B0, 2);
    MLP_SB();
    if (tn < nt2) fetch_q(B0, tn, tn_two, 0);
    MLP_SB();
    quarter(B1, 3);
    if (a.bias2) {
      ya = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_ya, one_k0, ya, 0, 0, 0);
      if (TWO) yb = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_yb, one_k0, yb, 0, 0, 0);
    }
    MLP_SB();
#ifdef XEQ_MLP_STAMPS
    MLP_STAMP(cx_, rx_); st2c_ += cx_ - cy_;
#endif
    if (row_ok) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned o = (unsigned)i * ldy32 + (unsigned)(32 * tt + 8 * g + 4 * kh);
        if (TWO) {
          *reinterpret_cast<float4*>(yo + o) = make_float4(ya[4 * g], ya[4 * g + 1], ya[4 * g + 2], ya[4 * g + 3]);
          *reinterpret_cast<float4*>(yo + o + 128u) = make_float4(yb[4 * g], yb[4 * g + 1], yb[4 * g + 2], yb[4 * g + 3]);
        } else {
          *reinterpret_cast<float4*>(yo + o) = make_float4(ya[4 * g], ya[4 * g + 1], ya[4 * g + 2], ya[4 * g + 3]);
        }
      }
    }
  };
  if (parts == 1) {
    for (; t0 + 4 < nt2; t0 += 8) pass(std::true_type{}, t0, t0 + 8, t0 + 12 < nt2);
    if (t0 < nt2) pass(std::false_type{}, t0, nt2, false);
  } else {
    for (; t0 < nt2; t0 += 4 * parts) pass(std::false_type{}, t0, t0 + 4 * parts, false);
  }
#ifdef XEQ_MLP_WGREC
  if (tid == 0 && blockIdx.x < 4096) {
    unsigned long long wg_t1_;
    unsigned hw, xcc;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(wg_t1_), "=s"(hw), "=s"(xcc)::"memory");
    g_mlp_wg[4 * blockIdx.x + 0] = hw;
    g_mlp_wg[4 * blockIdx.x + 1] = xcc;
    g_mlp_wg[4 * blockIdx.x + 2] = wg_t0_;
    g_mlp_wg[4 * blockIdx.x + 3] = wg_t1_;
  }
#endif
#ifdef XEQ_MLP_STAMPS
  MLP_STAMP(c1_, r1_);
  if (tid == 0) {
    atomicAdd(&g_mlp_stamps[0], c1_ - c0_);
    atomicAdd(&g_mlp_stamps[1], r1_ - r0_);
    atomicAdd(&g_mlp_stamps[2], 1ull);
    atomicAdd(&g_mlp_stamps[3], cp1_ - c0_);     // prologue: first fetches, first barrier
    atomicAdd(&g_mlp_stamps[4], cp2_ - cp1_);    // stage-1 chunk loop
    atomicAdd(&g_mlp_stamps[5], cp3_ - cp2_);    // hidden-tile epilogue + barrier
    atomicAdd(&g_mlp_stamps[6], st2c_);          // stage-2 MFMA quarters
    atomicAdd(&g_mlp_stamps[7], c1_ - cp3_);     // stage 2 in total
  }
#endif
}

// ---- 64 rows per workgroup -------------------------------------------------------------------------------------------
// The 32-row form reads every weight once per 32 rows: 360 KB per workgroup out of L2 for the message MLP, 207 MB per launch at
// 18 k nodes, and with every CU streaming at once the chip's L2 delivers ~7.5 TB/s (scratch/mfma_mix_probe.hip: eight 16-byte
// weight loads per 32 MFMAs slow the MFMA stream by 27 % at one or three waves per SIMD alike).  Two row tiles per workgroup share
// each weight fragment: half the weight traffic per MFMA, the matrix pipe is the bound again.  Same sums per row as the 32-row form
// (same k order, same even / odd accumulators in stage 1), so which form ran does not show in the results.  Used from one full
// round of 64-row tiles on (n >= 16 384); below that the 32-row form keeps more CUs busy.
template <bool REVERSE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) k_mlp2_r64(MlpArgs a) {
  constexpr int R = 64;
  __shared__ __attribute__((aligned(16))) float Xs[2][R * MLP_XLD];
  __shared__ __attribute__((aligned(16))) float Ts[R * MLP_TLD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kh = lane >> 5;
  int tile_, part_, parts_;
  a.ts.decode((int)blockIdx.x, tile_, part_, parts_);
  const int64_t row0 = (int64_t)tile_ * R;
  const int rows_here = (int)min((int64_t)R, a.n - row0);
  const int n_chunks = (a.K1 + MLP_CK - 1) / MLP_CK;
  const int g1 = a.K1 / 8;
  const int sr = tid >> 4, sc = (tid & 15) * 4;   // staging: 64 rows x 16 float4 per chunk, four per thread (rows sr + 16 k)
  const float* __restrict__ xb = a.X + row0 * a.ldx;
  float* __restrict__ preb = a.pre + row0 * MLP_H;
  float* __restrict__ yo = a.Y + row0 * a.ldy;
  const unsigned ldx32 = (unsigned)a.ldx, ldy32 = (unsigned)a.ldy;
  unsigned rc[4];
  bool rok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rc[k] = (unsigned)min(sr + 16 * k, rows_here - 1) * ldx32;
    rok[k] = sr + 16 * k < rows_here;
  }
  auto fetch = [&](int c, float4 (&v)[4]) {
    const int colc = min(MLP_CK * c + sc, a.K1 - 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(xb + (rc[k] + (unsigned)colc));
  };
  auto stash = [&](int c, const float4 (&v)[4]) {
    const bool cok = MLP_CK * c + sc < a.K1;
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(&Xs[c & 1][(sr + 16 * k) * MLP_XLD + sc]) = keep4(rok[k] && cok, v[k]);
  };
  const float4* w1 = reinterpret_cast<const float4*>(a.W1p) + (int64_t)wave * (g1 + 1) * 64;
  auto fetch_w = [&](float4 (&b)[4], int g0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = w1[min(g0 + q, g1 - 1) * 64 + lane];
  };
  f32x16 acc[2][2];   // [row tile][even / odd k steps]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f;
  }
  const float one_k0 = kh == 0 ? 1.f : 0.f;
  float bias_a = 0.f;
  if (a.bias1) bias_a = reinterpret_cast<const float*>(w1 + (int64_t)g1 * 64 + lane)[0];
  float4 pv[2][4];
  if (REVERSE) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pv[rt][g] = *reinterpret_cast<const float4*>(preb + ((unsigned)min(32 * rt + i, rows_here - 1) * MLP_H + (unsigned)(32 * wave + 8 * g + 4 * kh)));
  }
  float4 xv[4], B0[4], B1[4];
  fetch(0, xv);
  fetch_w(B0, 0);
  stash(0, xv);
  MLP_LDS_BARRIER();
  for (int c = 0; c < n_chunks; ++c) {
    const bool more = c + 1 < n_chunks;
    if (more) fetch(c + 1, xv);
    const float* xs = &Xs[c & 1][i * MLP_XLD + 4 * kh];
    auto half = [&](const float4 (&b)[4], int h) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x0 = *reinterpret_cast<const float4*>(xs + 8 * (4 * h + q));
        const float4 x1 = *reinterpret_cast<const float4*>(xs + 32 * MLP_XLD + 8 * (4 * h + q));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, x0.x, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, x1.x, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, x0.y, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, x1.y, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, x0.z, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, x1.z, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, x0.w, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, x1.w, acc[1][1], 0, 0, 0);
      }
    };
    fetch_w(B1, 8 * c + 4);
    MLP_SB();
    half(B0, 0);
    MLP_SB();
    fetch_w(B0, 8 * c + 8);
    MLP_SB();
    half(B1, 1);
    MLP_SB();
    if (more) stash(c + 1, xv);
    MLP_LDS_BARRIER();
  }
  if (a.bias1) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, one_k0, acc[0][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, one_k0, acc[1][0], 0, 0, 0);
  }
  const int nt2 = a.N2 / 32;
  constexpr int G2 = MLP_H / 8;
  const float4* w2 = reinterpret_cast<const float4*>(a.W2p);
  auto fetch_q = [&](float4 (&b)[4], int t, int qq) {
    const float4* wa = w2 + ((int64_t)t * (G2 + 1) + 4 * qq) * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = wa[q * 64 + lane];
  };
  const int parts = __builtin_amdgcn_readfirstlane(parts_), part = __builtin_amdgcn_readfirstlane(part_);
  int t0 = wave + 4 * part;
  if (t0 < nt2) fetch_q(B0, t0, 0);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = 32 * rt + i;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 32 * wave + 8 * g + 4 * kh;
      const float4 t = make_float4(acc[rt][0][4 * g] + acc[rt][1][4 * g], acc[rt][0][4 * g + 1] + acc[rt][1][4 * g + 1],
                                   acc[rt][0][4 * g + 2] + acc[rt][1][4 * g + 2], acc[rt][0][4 * g + 3] + acc[rt][1][4 * g + 3]);
      float4 v;
      if (!REVERSE) {
        if (row < rows_here && part_ == 0) *reinterpret_cast<float4*>(preb + ((unsigned)row * MLP_H + (unsigned)col)) = t;
        v = make_float4(silu_f(t.x), silu_f(t.y), silu_f(t.z), silu_f(t.w));
      } else {
        v = make_float4(t.x * silu_grad_f(pv[rt][g].x), t.y * silu_grad_f(pv[rt][g].y), t.z * silu_grad_f(pv[rt][g].z),
                        t.w * silu_grad_f(pv[rt][g].w));
      }
      *reinterpret_cast<float4*>(&Ts[row * MLP_TLD + col]) = v;
    }
  }
  MLP_LDS_BARRIER();
  const float* ts = &Ts[i * MLP_TLD + 4 * kh];
  for (; t0 < nt2; t0 += 4 * parts) {
    const int tn = t0 + 4 * parts;
    f32x16 ya, yb;   // the tile's rows 0..31 and 32..63
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ya[r] = 0.f;
      yb[r] = 0.f;
    }
    float bias_y = 0.f;
    if (a.bias2) bias_y = reinterpret_cast<const float*>(w2 + ((int64_t)t0 * (G2 + 1) + G2) * 64 + lane)[0];
    auto quarter = [&](const float4 (&b)[4], int qq) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 u0 = *reinterpret_cast<const float4*>(ts + 8 * (4 * qq + q));
        const float4 u1 = *reinterpret_cast<const float4*>(ts + 32 * MLP_TLD + 8 * (4 * qq + q));
        ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, u0.x, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].x, u1.x, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, u0.y, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].y, u1.y, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, u0.z, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].z, u1.z, yb, 0, 0, 0);
        ya = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, u0.w, ya, 0, 0, 0);
        yb = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q].w, u1.w, yb, 0, 0, 0);
      }
    };
    fetch_q(B1, t0, 1);
    MLP_SB();
    quarter(B0, 0);
    MLP_SB();
    fetch_q(B0, t0, 2);
    MLP_SB();
    quarter(B1, 1);
    MLP_SB();
    fetch_q(B1, t0, 3);
    MLP_SB();
    quarter(B0, 2);
    MLP_SB();
    if (tn < nt2) fetch_q(B0, tn, 0);
    MLP_SB();
    quarter(B1, 3);
    if (a.bias2) {
      ya = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_y, one_k0, ya, 0, 0, 0);
      yb = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_y, one_k0, yb, 0, 0, 0);
    }
    MLP_SB();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned o = (unsigned)i * ldy32 + (unsigned)(32 * t0 + 8 * g + 4 * kh);
      if (i < rows_here) *reinterpret_cast<float4*>(yo + o) = make_float4(ya[4 * g], ya[4 * g + 1], ya[4 * g + 2], ya[4 * g + 3]);
      if (32 + i < rows_here) *reinterpret_cast<float4*>(yo + o + 32u * ldy32) = make_float4(yb[4 * g], yb[4 * g + 1], yb[4 * g + 2], yb[4 * g + 3]);
    }
  }
}

// ---- few rows (MD-sized systems) ------------------------------------------------------------------------------------------------
// 16 x 16 exact-f32 tiles (v_mfma_f32_16x16x4_f32): the same fused-multiply-add chain per output element as the 32-row form -- the
// instruction rounds like a sequential fmaf chain in k order whatever its shape (xeq_linear.hip, scratch/mfma_order) -- so the results
// are BIT-EQUAL, with a quarter of the chain per wave and four times the waves.  A workgroup of 8 waves owns 16 rows: wave w forms
// hidden columns [16 w, 16 w + 16) (stage 1: the even / odd accumulators of k_mlp2, one instruction per k-group each, summed at the
// end), then output tile 8 part + w (stage 2, k_mlp2's single chain); `parts` workgroups share a row tile, each redoing stage 1.
// Weights from the same packed copies; stage 2's fragments and the first 128 k of stage 1's are requested before anything else.
constexpr int MLP_S_ROWS = 16;

template <bool REVERSE, int NCH>   // NCH: chunks of 16 k-groups (128 k) in stage 1
__device__ __forceinline__ void mlp2_s_body(const MlpArgs& a, int parts, float* mlp_s_lds, int block, int tid) {
  const int XLD = a.K1 + 4;
  float* Xs = mlp_s_lds;                      // [16][K1 + 4]
  float* Ts = Xs + MLP_S_ROWS * XLD;          // [16][MLP_TLD]
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4, kh = kq & 1;
  const int sel = kq >> 1;
  const int rt = block / parts, part = block - rt * parts;
  const int64_t row0 = (int64_t)rt * MLP_S_ROWS;
  const int rows_here = (int)min((int64_t)MLP_S_ROWS, a.n - row0);
  const int g1 = a.K1 >> 3;
  constexpr int G2 = MLP_H / 8, G1 = 16 * NCH;
  // Requests in the order of use -- a wave's loads return in order: the rows first (16 rows x K1: at most NCH float4 per thread, held in
  // registers until they go to LDS), then EVERY weight fragment (the chains then wait for memory once, and start while the tail of
  // the weights is still on its way): stage 1 whole -- a lane takes the two components of its float4 it multiplies (x, y: even / odd
  // accumulator of lanes kq < 2; z, w: of lanes kq >= 2) as ONE 8-byte load -- then the first stage-2 tile.
  const int k4 = a.K1 >> 2;
  float4 xr[NCH];
  int xo[NCH];   // LDS offset of the float4, -1: none
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int idx = tid + 512 * j;
    const int r = idx / k4, c4 = idx - r * k4;
    const bool in = idx < MLP_S_ROWS * k4;
    const float4 v = *reinterpret_cast<const float4*>(a.X + (row0 + min(r, rows_here - 1)) * a.ldx + 4 * (in ? c4 : 0));
    xr[j] = keep4(in && r < rows_here, v);
    xo[j] = in ? r * XLD + 4 * c4 : -1;
  }
  const unsigned ic = (unsigned)min(i, rows_here - 1);
  float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (REVERSE) pv = *reinterpret_cast<const float4*>(a.pre + (row0 + ic) * MLP_H + 16 * wave + 4 * kq);
  MLP_SB();
  const float2* w1 = reinterpret_cast<const float2*>(reinterpret_cast<const float4*>(a.W1p) + (int64_t)(wave >> 1) * (g1 + 1) * 64 +
                                                     16 * (wave & 1) + i + 32 * kh) + sel;
  float2 eo[G1];
#pragma unroll
  for (int q = 0; q < G1; ++q) eo[q] = w1[(q < g1 ? q : g1 - 1) * 128];
  const float bias1_a = (a.bias1 && kq == 0) ? reinterpret_cast<const float*>(w1 + (int64_t)g1 * 128)[0] : 0.f;
  const int nt16 = a.N2 >> 4;
  int t2 = 8 * part + wave;
  float wb[G2][2];
  float bias2_a = 0.f;
  auto fetch2 = [&](int t) {
    const int tc = t < nt16 ? t : 0;
    const float4* w2 = reinterpret_cast<const float4*>(a.W2p) + (int64_t)(tc >> 1) * (G2 + 1) * 64 + 16 * (tc & 1) + i + 32 * kh;
#pragma unroll
    for (int q = 0; q < G2; ++q) {
      const float4 v = w2[q * 64];
      wb[q][0] = sel ? v.y : v.x;
      wb[q][1] = sel ? v.w : v.z;
    }
    bias2_a = (a.bias2 && kq == 0) ? reinterpret_cast<const float*>(w2 + (int64_t)G2 * 64)[0] : 0.f;
  };
  MLP_SB();
  // the rows into LDS (the weights stay in flight across the barrier: it orders LDS traffic only)
#pragma unroll
  for (int j = 0; j < NCH; ++j)
    if (xo[j] >= 0) *reinterpret_cast<float4*>(&Xs[xo[j]]) = xr[j];
  MLP_LDS_BARRIER();
  fetch2(t2);   // (behind the stage-1 weights in any case; here the rows' registers are free again)
  MLP_SB();
  // ---- stage 1: hidden columns [16 wave, 16 wave + 16)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* xs = &Xs[i * XLD + 4 * kh + 2 * sel];
#pragma unroll
  for (int q = 0; q < G1; ++q)
    if (q < g1) {
      const float2 xv = *reinterpret_cast<const float2*>(xs + 8 * q);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(eo[q].x, xv.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(eo[q].y, xv.y, acc1, 0, 0, 0);
    }
  if (a.bias1) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bias1_a, kq == 0 ? 1.f : 0.f, acc0, 0, 0, 0);
  {
    const int col = 16 * wave + 4 * kq;
    const float4 t = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
    float4 v;
    if (!REVERSE) {
      if (i < rows_here && part == 0) *reinterpret_cast<float4*>(a.pre + (row0 + i) * MLP_H + col) = t;
      v = make_float4(silu_f(t.x), silu_f(t.y), silu_f(t.z), silu_f(t.w));
    } else {
      v = make_float4(t.x * silu_grad_f(pv.x), t.y * silu_grad_f(pv.y), t.z * silu_grad_f(pv.z), t.w * silu_grad_f(pv.w));
    }
    *reinterpret_cast<float4*>(&Ts[i * MLP_TLD + col]) = v;
  }
  MLP_LDS_BARRIER();
  // ---- stage 2: output tiles 8 part + wave, + 8 parts, ...
  const float* ts = &Ts[i * MLP_TLD + 4 * kh];
  for (; t2 < nt16; t2 += 8 * parts) {
    f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < G2; ++q) {
      const float4 tv = *reinterpret_cast<const float4*>(ts + 8 * q);
      y = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[q][0], sel ? tv.y : tv.x, y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[q][1], sel ? tv.w : tv.z, y, 0, 0, 0);
    }
    if (a.bias2) y = __builtin_amdgcn_mfma_f32_16x16x4f32(bias2_a, kq == 0 ? 1.f : 0.f, y, 0, 0, 0);
    if (i < rows_here) *reinterpret_cast<float4*>(a.Y + (row0 + i) * a.ldy + 16 * t2 + 4 * kq) = make_float4(y[0], y[1], y[2], y[3]);
    if (t2 + 8 * parts < nt16) fetch2(t2 + 8 * parts);
  }
}

template <bool REVERSE, int NCH>
__global__ void __launch_bounds__(512) k_mlp2_s(MlpArgs a, int parts) {
  extern __shared__ __attribute__((aligned(16))) float mlp_s_lds[];
  mlp2_s_body<REVERSE, NCH>(a, parts, mlp_s_lds, (int)blockIdx.x, (int)threadIdx.x);
}

// Two INDEPENDENT products of a node block in one launch (a captured MD-sized step pays ~5 us per launch whatever it does): the first
// n_mlp workgroups are k_mlp2_s, the rest k_linear_s with eight waves.  XPainnUpdate.forward: a = update_mlp([shat | v]) beside
// <U, V> -> dot_lin (nn/xpainn.py:219-223); its reverse: dL/d[shat | v] beside dL/dp.  Each half computes what its own launch would.
template <bool REVERSE, int NCH>
__global__ void __launch_bounds__(512) k_mlp2_linear_s(MlpArgs a, int parts, int n_mlp, LinArgs b) {
  extern __shared__ __attribute__((aligned(16))) float mlp_s_lds[];
  if ((int)blockIdx.x < n_mlp) mlp2_s_body<REVERSE, NCH>(a, parts, mlp_s_lds, (int)blockIdx.x, (int)threadIdx.x);
  else linear_s_body<8>(b, mlp_s_lds, (int)blockIdx.x - n_mlp, (int)threadIdx.x);
}

constexpr int MLP_S_KMAX = 640;   // five chunks: 160 weight registers in stage 1

template <bool REVERSE>
static void mlp_launch_small(const MlpArgs& a, hipStream_t stream, const LinArgs* lin = nullptr) {
  // workgroups per row tile: one output tile per wave while the chip has idle CUs (every workgroup redoes stage 1)
  const int64_t row_tiles = (a.n + MLP_S_ROWS - 1) / MLP_S_ROWS;
  const int max_parts = (a.N2 / 16 + 7) / 8;
  int parts = (int)(256 / row_tiles);
  parts = parts < 1 ? 1 : (parts > max_parts ? max_parts : parts);
  size_t shmem = sizeof(float) * ((size_t)MLP_S_ROWS * (a.K1 + 4) + (size_t)MLP_S_ROWS * MLP_TLD);
  const int n_mlp = (int)row_tiles * parts;
  const int nch = (a.K1 / 8 + 15) / 16;
  if (lin) {
    const size_t lin_shmem = sizeof(float) * (size_t)LIN_S_ROWS * LIN_XLD;
    if (lin_shmem > shmem) shmem = lin_shmem;
    const int n_lin = (int)((lin->n + LIN_S_ROWS - 1) / LIN_S_ROWS) * ((lin->n_out + 127) / 128);
    const dim3 grid((unsigned)(n_mlp + n_lin));
    if (nch <= 1) hipLaunchKernelGGL((k_mlp2_linear_s<REVERSE, 1>), grid, dim3(512), shmem, stream, a, parts, n_mlp, *lin);
    else if (nch <= 3) hipLaunchKernelGGL((k_mlp2_linear_s<REVERSE, 3>), grid, dim3(512), shmem, stream, a, parts, n_mlp, *lin);
    else hipLaunchKernelGGL((k_mlp2_linear_s<REVERSE, 5>), grid, dim3(512), shmem, stream, a, parts, n_mlp, *lin);
    return;
  }
  const dim3 grid((unsigned)n_mlp);
  if (nch <= 1) hipLaunchKernelGGL((k_mlp2_s<REVERSE, 1>), grid, dim3(512), shmem, stream, a, parts);
  else if (nch <= 3) hipLaunchKernelGGL((k_mlp2_s<REVERSE, 3>), grid, dim3(512), shmem, stream, a, parts);
  else hipLaunchKernelGGL((k_mlp2_s<REVERSE, 5>), grid, dim3(512), shmem, stream, a, parts);
}

constexpr int64_t MLP_R64_MIN_ROWS = 64 * 256;   // one full round of 64-row tiles
// ... and an output wide enough (>= 4 groups of four tiles) that a short last round can be split: measured at 18 k nodes, 64-row
// against 32-row form: 576 outputs 46.6 / 49.6 us, 480: 56.5 / 56.5, 352: 60.1 / 55.8, 128: 57.5 / 49.3; at 147 k nodes 280 / 332
// (576 outputs) and 278 / 278 (128 outputs: the reverse forms are not bound by their weight stream)
static bool mlp_use_r64(int64_t n, int n2) { return n >= MLP_R64_MIN_ROWS && (n2 / 32 + 3) / 4 >= 4; }

static int mlp_check(const char* name, int64_t n, int k1, int n2, int64_t ldx, int64_t ldy) {
  XEQ_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) * MLP_ROWS, "%s: n = %lld out of range", name, (long long)n);
  XEQ_CHECK_ARG(k1 > 0 && k1 % 8 == 0 && n2 > 0 && n2 % 32 == 0, "%s: needs k1 %% 8 == 0 and n2 %% 32 == 0 (got %d, %d)", name, k1, n2);
  XEQ_CHECK_ARG(ldx >= k1 && ldx % 4 == 0 && ldy >= n2 && ldy % 4 == 0, "%s: row strides (%lld, %lld) do not fit (%d, %d) / 16-byte rows", name,
                (long long)ldx, (long long)ldy, k1, n2);
  return XEQ_OK;
}

}  // namespace xeq

using namespace xeq;

extern "C" {

int xeq_mlp_debug_stamps(unsigned long long out[8]) {   // development only (see XEQ_MLP_STAMPS); reads and clears
  unsigned long long zero[8] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_stamps), sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_stamps), zero, sizeof(zero)) != hipSuccess) return XEQ_ERR_LAUNCH;
  return XEQ_OK;
}

int xeq_mlp_debug_wg(unsigned long long* out) {   // development only: 4096 x 4
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_wg), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? XEQ_OK : XEQ_ERR_LAUNCH;
}

/* How a launch of the node kernels cuts `tiles` 32-node tiles into workgroups (TileSplit, xeq_common.h): out = {tiles run by one
 * workgroup each, workgroups per remaining tile, grid size}.  Host logic only; what tests/test_host_logic.py checks. */
int xeq_node_tile_split(int64_t tiles, int max_split, int64_t out[3]) {
  XEQ_CHECK_ARG(tiles >= 0 && tiles < ((int64_t)1 << 31) && out, "xeq_node_tile_split: bad arguments");
  const TileSplit t = tile_split(tiles, max_split);
  out[0] = t.n_full;
  out[1] = t.split;
  out[2] = t.grid(tiles);
  return XEQ_OK;
}

int xeq_mlp2_supported(int dtype, int k1, int hidden, int n2) {
  return dtype == XEQ_F32 && hidden == MLP_H && k1 > 0 && k1 % 32 == 0 && n2 > 0 && n2 % 32 == 0;   // k1 is the reverse pass's n2
}

int64_t xeq_mlp_packed_floats(int n_out, int k_in) { return (int64_t)(n_out / 32) * (k_in / 8 + 1) * 256; }

int xeq_mlp_pack(const float* w, const float* bias, int n_out, int k_in, int transposed, float* out, void* stream) {
  XEQ_CHECK_ARG(w && out, "xeq_mlp_pack: null buffer");
  XEQ_CHECK_ARG(n_out > 0 && n_out % 32 == 0 && k_in > 0 && k_in % 8 == 0, "xeq_mlp_pack: needs n_out %% 32 == 0 and k_in %% 8 == 0 (got %d, %d)",
                n_out, k_in);
  const int64_t total = xeq_mlp_packed_floats(n_out, k_in) / 4;
  hipLaunchKernelGGL(k_mlp_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, bias, n_out, k_in, transposed, out);
  XEQ_CHECK_LAUNCH("xeq_mlp_pack");
  return XEQ_OK;
}

int xeq_mlp2_fwd(const float* x, int64_t ldx, int64_t n, int k1, const float* w1p, const float* w2p, int n2, float* pre, float* y,
                 int64_t ldy, void* stream) {
  if (int rc = mlp_check("xeq_mlp2_fwd", n, k1, n2, ldx, ldy)) return rc;
  XEQ_CHECK_ARG(n == 0 || (x && w1p && w2p && pre && y), "xeq_mlp2_fwd: null buffer");
  if (n == 0) return XEQ_OK;
  const bool r64 = mlp_use_r64(n, n2);
  const int64_t tiles = r64 ? (n + 63) / 64 : (n + MLP_ROWS - 1) / MLP_ROWS;
  MlpArgs a{x, ldx, n, k1, n2, w1p, w2p, 1, 1, pre, y, ldy, tile_split(tiles, (n2 / 32 + 3) / 4)};
  if (n <= xeq_small_rows() && k1 <= MLP_S_KMAX) mlp_launch_small<false>(a, (hipStream_t)stream);
  else if (r64) hipLaunchKernelGGL(k_mlp2_r64<false>, dim3(a.ts.grid(tiles)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_mlp2<false>, dim3(a.ts.grid(tiles)), dim3(256), 0, (hipStream_t)stream, a);
  XEQ_CHECK_LAUNCH("xeq_mlp2_fwd");
  return XEQ_OK;
}

int xeq_mlp2_bwd(const float* g, int64_t ldg, int64_t n, int k1, const float* w2tp, const float* pre, const float* w1tp, int n2,
                 float* gx, int64_t ldgx, void* stream) {
  if (int rc = mlp_check("xeq_mlp2_bwd", n, k1, n2, ldg, ldgx)) return rc;
  XEQ_CHECK_ARG(n == 0 || (g && w2tp && w1tp && pre && gx), "xeq_mlp2_bwd: null buffer");
  if (n == 0) return XEQ_OK;
  const bool r64 = mlp_use_r64(n, n2);
  const int64_t tiles = r64 ? (n + 63) / 64 : (n + MLP_ROWS - 1) / MLP_ROWS;
  MlpArgs a{g, ldg, n, k1, n2, w2tp, w1tp, 0, 0, const_cast<float*>(pre), gx, ldgx, tile_split(tiles, (n2 / 32 + 3) / 4)};
  if (n <= xeq_small_rows() && k1 <= MLP_S_KMAX) mlp_launch_small<true>(a, (hipStream_t)stream);
  else if (r64) hipLaunchKernelGGL(k_mlp2_r64<true>, dim3(a.ts.grid(tiles)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_mlp2<true>, dim3(a.ts.grid(tiles)), dim3(256), 0, (hipStream_t)stream, a);
  XEQ_CHECK_LAUNCH("xeq_mlp2_bwd");
  return XEQ_OK;
}

/* XPainnUpdate's two independent products side by side (nn/xpainn.py:219-223 and their reverse): the two-layer MLP of xeq_mlp2_fwd /
 * _bwd and the single linear layer of xeq_linear_fwd (no bias, no activation, no row gather) -- ONE launch when the rows take the
 * few-row forms, else the two launches one after the other.  Results are those of the separate entry points bit for bit. */
int xeq_mlp2_and_linear(int reverse, const float* x, int64_t ldx, int64_t n, int k1, const float* w1p, const float* w2p, int n2, float* pre,
                        float* y, int64_t ldy, const float* lin_x, int64_t lin_ldx, int lin_k, const float* lin_wp, int lin_n_out,
                        float* lin_y, int64_t lin_ldy, void* stream) {
  const bool one = n > 0 && n <= xeq_small_rows() && k1 <= MLP_S_KMAX && xeq_linear_supported(XEQ_F32, lin_k, lin_n_out) &&
                   lin_ldx % 4 == 0 && lin_ldy % 4 == 0 && lin_ldx >= lin_k && lin_ldy >= lin_n_out && lin_x && lin_wp && lin_y;
  if (!one) {
    const int rc = reverse ? xeq_mlp2_bwd(x, ldx, n, k1, w1p, pre, w2p, n2, y, ldy, stream)
                           : xeq_mlp2_fwd(x, ldx, n, k1, w1p, w2p, n2, pre, y, ldy, stream);
    if (rc != XEQ_OK) return rc;
    return xeq_linear_fwd(lin_x, lin_ldx, n, lin_k, nullptr, lin_wp, lin_n_out, 0, 0, nullptr, lin_y, lin_ldy, stream);
  }
  if (int rc = mlp_check("xeq_mlp2_and_linear", n, k1, n2, ldx, ldy)) return rc;
  XEQ_CHECK_ARG(x && w1p && w2p && pre && y, "xeq_mlp2_and_linear: null buffer");
  MlpArgs a{x, ldx, n, k1, n2, w1p, w2p, reverse ? 0 : 1, reverse ? 0 : 1, pre, y, ldy, tile_split(1, 1)};
  LinArgs b{lin_x, nullptr, lin_ldx, n, lin_k, lin_n_out, lin_wp, 0, 0, nullptr, lin_y, lin_ldy};
  if (reverse) mlp_launch_small<true>(a, (hipStream_t)stream, &b);
  else mlp_launch_small<false>(a, (hipStream_t)stream, &b);
  XEQ_CHECK_LAUNCH("xeq_mlp2_and_linear");
  return XEQ_OK;
}

}  // extern "C"
