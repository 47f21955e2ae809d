// The few-row form of the single linear layers (k_linear_s, csrc/xeq_linear.hip) as a device function: also one half of the launches
// that run two independent products of a node block side by side (csrc/xeq_mlp.hip: k_mlp2_linear_s).
#pragma once
#include "xeq_common.h"

namespace xeq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LIN_KMAX = 256;
constexpr int LIN_XLD = LIN_KMAX + 4;
constexpr int LIN_S_ROWS = 16;

struct LinArgs {
  const float* X;           // [*, ldx]
  const int32_t* row_index; // optional: row r of the operand is X[row_index[r]]
  int64_t ldx, n;
  int K, n_out;
  const float* Wp;          // packed [n_out / 32][K / 8 + 1][64][4]
  int has_bias, act;        // act: 0 none, 1 SiLU
  float* pre;               // optional [n, n_out]: the pre-activation (what the reverse pass of a SiLU layer needs)
  float* Y;                 // [n, ldy]
  int64_t ldy;
};

__device__ __forceinline__ float lin_silu(float x) { return x / (1.f + expf(-x)); }

// ---- the same product for FEW rows (MD-sized systems: 21 .. a few thousand atoms) ---------------------------------------------------------
// An exact-f32 matrix instruction is a chain of fused multiply-adds in k order, whatever its tile shape: v_mfma_f32_16x16x4_f32 fed the
// k sequence of k_linear's 32x32x2 tiles gives the SAME BITS (scratch/mfma_order/order.hip: 0 of 51 200 outputs differ, and both equal a
// sequential fmaf chain).  So when the rows do not fill the chip a tile is 16 rows x 16 columns: a quarter of the chain per wave
// (K / 4 instructions of 32 cycles against K / 2 of 64) and four times the waves, with results bit-equal to the large form -- the
// row count decides the form, never the result.  A workgroup (4 waves) owns 16 rows x 64 columns; the weight fragments come from the
// SAME packed copy (lane (i, kq) of a 16x16x4 instruction takes k = 8 q + 4 (kq & 1) + 2 s + (kq >> 1), s = 0, 1: the order in which
// the large form's four instructions of a k-group visit the eight k), ALL of a tile's fragments are requested before the first
// instruction (K <= 256: 64 registers), so the chain waits for memory once.
// NW: waves per workgroup (16 rows x 16 NW columns each); Xs: [16][LIN_XLD] floats of LDS; block: the workgroup's index among this
// product's ((n + 15) / 16) * ((n_out + 16 NW - 1) / (16 NW)) workgroups.
template <int NW>
__device__ __forceinline__ void linear_s_body(const LinArgs& a, float* Xs, int block, int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4, kh = kq & 1;
  const bool sel = (kq >> 1) != 0;
  const int cgs = (a.n_out + 16 * NW - 1) / (16 * NW);   // column groups; the last may hold fewer tiles (n_out a multiple of 32)
  const int rt = block / cgs, cg = block - rt * cgs;
  const int64_t row0 = (int64_t)rt * LIN_S_ROWS;
  const int rows_here = (int)min((int64_t)LIN_S_ROWS, a.n - row0);
  const int G = a.K >> 3;
  const bool tile_ok = 16 * (NW * cg + wave) < a.n_out;
  const int t16 = tile_ok ? NW * cg + wave : 0;   // (a wave without a tile works on tile 0 and stores nothing)
  // requests in the order of use (a wave's loads return in order): the rows (K <= 256: 1024 / (64 NW) float4 per thread, in registers
  // until they go to LDS), then all of the tile's weight fragments
  constexpr int XN = 1024 / (64 * NW);
  const int k4 = a.K >> 2;
  float4 xr[XN];
  int xo[XN];
#pragma unroll
  for (int j = 0; j < XN; ++j) {
    const int idx = tid + 64 * NW * j;
    const int r = idx / k4, c4 = idx - r * k4;
    const bool in = idx < LIN_S_ROWS * k4;
    int64_t src = row0 + min(r, rows_here - 1);
    if (a.row_index) src = a.row_index[src];
    const float4 v = *reinterpret_cast<const float4*>(a.X + src * a.ldx + 4 * (in ? c4 : 0));
    const bool ok = in && r < rows_here;
    xr[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    xo[j] = in ? r * LIN_XLD + 4 * c4 : -1;
  }
  __builtin_amdgcn_sched_barrier(0);
  const float4* wp = reinterpret_cast<const float4*>(a.Wp) + (int64_t)(t16 >> 1) * (G + 1) * 64 + 16 * (t16 & 1) + i + 32 * kh;
  float wa[LIN_KMAX / 8][2];
#pragma unroll
  for (int q = 0; q < LIN_KMAX / 8; ++q) {
    const float4 v = wp[(q < G ? q : G - 1) * 64];
    wa[q][0] = sel ? v.y : v.x;
    wa[q][1] = sel ? v.w : v.z;
  }
  const float bias_a = (a.has_bias && kq == 0) ? reinterpret_cast<const float*>(wp + (int64_t)G * 64)[0] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < XN; ++j)
    if (xo[j] >= 0) *reinterpret_cast<float4*>(&Xs[xo[j]]) = xr[j];
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // orders LDS traffic only: the weights stay in flight
  const float* xs = &Xs[i * LIN_XLD + 4 * kh];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < LIN_KMAX / 8; ++q)
    if (q < G) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + 8 * q);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][0], sel ? xv.y : xv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][1], sel ? xv.w : xv.z, acc, 0, 0, 0);
    }
  if (a.has_bias) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bias_a, kq == 0 ? 1.f : 0.f, acc, 0, 0, 0);
  if (i < rows_here && tile_ok) {
    const int64_t row = row0 + i;
    const int col = 16 * t16 + 4 * kq;
    float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (a.pre) *reinterpret_cast<float4*>(a.pre + row * a.n_out + col) = v;
    if (a.act == 1) v = make_float4(lin_silu(v.x), lin_silu(v.y), lin_silu(v.z), lin_silu(v.w));
    *reinterpret_cast<float4*>(a.Y + row * a.ldy + col) = v;
  }
}

}  // namespace xeq
