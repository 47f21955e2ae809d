// Dense node-side contractions of the path (scalar_mlp, update_mlp, o3.Linear U|V, dot_lin
// and their reverse products): C[M, N] = A[M, K] x B[K, N] (+ bias, + SiLU) in exact f32 on the matrix cores.
//
// The shapes are tall and skinny -- M = nodes (x 2l+1), K and N in 32 .. 576 -- so the operand that matters is A / C
// (streamed once); B is a weight matrix of at most 1.3 MB that every wave re-reads out of L2.  No LDS, no barriers:
//   * a wave owns RW x 32 rows and NBW x 32 columns; v_mfma_f32_32x32x2_f32 wants, per lane (i = lane & 31,
//     h = lane >> 5), A[row i][k] and B[k][col i] for k = 2 s + h.  The sum over k is order-free, so the K axis is
//     walked in chunks of 8 with k = 8 t + 4 h + j for step j: then a lane's four A values of a chunk are ONE 16-byte
//     load from its row, and (for weights stored [N, K], i.e. x @ W^T) its four B values are one 16-byte load too;
//     weights stored [K, N] are four dword loads, coalesced over the 32 columns;
//   * chunks are 8 PD values of k deep (PD = 4 where K allows: a 128-byte line of every row per chunk) and chunk
//     t + 1 is loaded while chunk t is multiplied (register double buffer): 16 RW NBW MFMAs hide the loads;
//   * the accumulator layout (row = (r & 3) + 8 (r >> 2) + 4 h, col = i) stores 128 contiguous bytes per register.
// The epilogue adds the bias and optionally applies SiLU, writing the pre-activation beside it (the reverse pass
// needs it), which removes the separate activation launch of the reference's Linear -> SiLU -> Linear stacks.
#include "xeq_common.h"

namespace xeq {

typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef float f32x16g __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  float* Cpre;
  int64_t M, lda, ldb, ldc;
  int K, N, nb_total, col_groups;
  int act;  // 0: none, 1: SiLU (C = silu(z), Cpre = z when given)
};

// One chunk = KC = 8 PD values of k; half h of the wave takes the contiguous half [KC/2 h, KC/2 (h + 1)) of it, so a
// lane's A values of a chunk are PD consecutive 16-byte loads (with both halves: 32 PD contiguous bytes of its row).
// bp points at B[k = KC t + KC/2 h][col] (layout [K, N]) or B[col][KC t + KC/2 h] ([N, K]).
template <bool B_NK, int PD>
__device__ __forceinline__ void gemm_load_b(const float* bp, int64_t ldb, float (&out)[PD][4]) {
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    if constexpr (B_NK) {
      const f32x4g v = *reinterpret_cast<const f32x4g*>(bp + 4 * p);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[p][j] = v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[p][j] = bp[(4 * p + j) * ldb];
    }
  }
}

#ifndef XEQ_GEMM_WPE
#define XEQ_GEMM_WPE(RW, NBW, PD) ((RW) * (NBW) * (PD) >= 16 ? 2 : ((RW) * (NBW) * (PD) >= 8 ? 3 : 4))
#endif

template <int RW, int NBW, bool B_NK, int PD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XEQ_GEMM_WPE(RW, NBW, PD)))) k_gemm_f32(GemmArgs a) {
  constexpr int KC = 8 * PD;
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  // consecutive waves take consecutive row tiles of one column group: the four waves of a workgroup share B lines
  const int64_t row_tiles = (a.M + 32 * RW - 1) / (32 * RW);
  const int64_t cg = wave / row_tiles, rt = wave - cg * row_tiles;
  if (cg >= a.col_groups) return;
  const int64_t r0 = rt * (32 * RW);
  const int nb0 = (int)cg * NBW;

  const float* ap[RW];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw) {
    int64_t r = r0 + 32 * rw + i;
    if (r >= a.M) r = a.M - 1;
    ap[rw] = a.A + r * a.lda + (KC / 2) * h;
  }
  const float* bp[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    int col = 32 * (nb0 + nb) + i;
    if (col >= a.N) col = a.N - 1;   // masked columns compute values that are never stored
    bp[nb] = B_NK ? a.B + (int64_t)col * a.ldb + (KC / 2) * h : a.B + (int64_t)((KC / 2) * h) * a.ldb + col;
  }
  const int64_t bstep = B_NK ? KC : KC * a.ldb;
  f32x16g acc[RW][NBW];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[rw][nb][q] = 0.f;

  const int nt = a.K / KC;
  f32x4g a0[RW][PD], a1[RW][PD];
  float b0[NBW][PD][4], b1[NBW][PD][4];
  auto load = [&](f32x4g (&av)[RW][PD], float (&bv)[NBW][PD][4]) {   // the operands at ap / bp, which then advance
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
#pragma unroll
      for (int p = 0; p < PD; ++p) av[rw][p] = *reinterpret_cast<const f32x4g*>(ap[rw] + 4 * p);
      ap[rw] += KC;
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      gemm_load_b<B_NK, PD>(bp[nb], a.ldb, bv[nb]);
      bp[nb] += bstep;
    }
  };
  auto multiply = [&](const f32x4g (&av)[RW][PD], const float (&bv)[NBW][PD][4]) {
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
            acc[rw][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rw][p][j], bv[nb][p][j], acc[rw][nb], 0, 0, 0);
  };
  // Ping-pong between two register buffers with NO conditional load inside the loop: a conditional load would merge a
  // path without the newer loads into the loop head, and the waits in front of the MFMAs would then have to drain
  // the prefetch as well (vmcnt is an in-order count).
  load(a0, b0);
  int t = 0;
  for (; t + 2 < nt; t += 2) {
    load(a1, b1);
    multiply(a0, b0);
    load(a0, b0);
    multiply(a1, b1);
  }
  if (t + 1 < nt) {
    load(a1, b1);
    multiply(a0, b0);
    multiply(a1, b1);
  } else {
    multiply(a0, b0);
  }

  // ---- epilogue: register q of the lane is row (q & 3) + 8 (q >> 2) + 4 h, column i of the 32 x 32 block
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int col = 32 * (nb0 + nb) + i;
    if (col >= a.N) continue;
    const float bz = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t r = r0 + 32 * rw + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (r >= a.M) continue;
        const float z = acc[rw][nb][q] + bz;
        if (a.act == 1) {
          if (a.Cpre) a.Cpre[r * a.ldc + col] = z;
          a.C[r * a.ldc + col] = z / (1.f + __expf(-z));
        } else {
          a.C[r * a.ldc + col] = z;
        }
      }
    }
  }
}

}  // namespace xeq

using namespace xeq;

template <int RW, int NBW>
static void gemm_launch(const GemmArgs& a, int b_layout, hipStream_t stream) {
  GemmArgs g = a;
  g.col_groups = (a.nb_total + NBW - 1) / NBW;
  const int64_t row_tiles = (a.M + 32 * RW - 1) / (32 * RW);
  const int64_t waves = row_tiles * g.col_groups;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  // chunk depth: 32 values of k where K allows it (a full 128-byte line of every row per chunk), else 16 or 8
  const int pd = (a.K % 32 == 0 && a.K >= 64) ? 4 : ((a.K % 16 == 0 && a.K >= 32) ? 2 : 1);
#define XEQ_GEMM_GO(PD_)                                                                                    \
  do {                                                                                                      \
    if (b_layout == 1)                                                                                      \
      hipLaunchKernelGGL((k_gemm_f32<RW, NBW, true, PD_>), dim3(grid), dim3(256), 0, stream, g);            \
    else                                                                                                    \
      hipLaunchKernelGGL((k_gemm_f32<RW, NBW, false, PD_>), dim3(grid), dim3(256), 0, stream, g);           \
  } while (0)
  if (pd == 4) XEQ_GEMM_GO(4);
  else if (pd == 2) XEQ_GEMM_GO(2);
  else XEQ_GEMM_GO(1);
#undef XEQ_GEMM_GO
}

extern "C" {

int xeq_gemm_f32(const float* A, int64_t M, int K, int64_t lda, const float* B, int b_layout, int64_t ldb, int N,
                 const float* bias, int act, float* C, int64_t ldc, float* C_pre, int tile, void* stream) {
  XEQ_CHECK_ARG(M >= 0 && K > 0 && N > 0, "xeq_gemm_f32: bad sizes");
  XEQ_CHECK_ARG((K & 7) == 0, "xeq_gemm_f32: K must be a multiple of 8 (got %d)", K);
  XEQ_CHECK_ARG((lda & 3) == 0 && lda >= K && ldc >= N, "xeq_gemm_f32: lda must be a multiple of 4 and cover K; ldc must cover N");
  XEQ_CHECK_ARG(b_layout == 0 || b_layout == 1, "xeq_gemm_f32: b_layout is 0 ([K, N]) or 1 ([N, K])");
  XEQ_CHECK_ARG(b_layout == 0 ? ldb >= N : (ldb >= K && (ldb & 3) == 0), "xeq_gemm_f32: bad ldb");
  XEQ_CHECK_ARG(act == 0 || act == 1, "xeq_gemm_f32: act is 0 (none) or 1 (SiLU)");
  XEQ_CHECK_ARG((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (b_layout == 0 || (reinterpret_cast<uintptr_t>(B) & 15) == 0),
                "xeq_gemm_f32: A (and B in the [N, K] layout) must be 16-byte aligned");
  if (M == 0) return XEQ_OK;
  GemmArgs a{};
  a.A = A; a.B = B; a.bias = bias; a.C = C; a.Cpre = C_pre;
  a.M = M; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.K = K; a.N = N; a.nb_total = (N + 31) / 32; a.act = act;
  // tile = 10 RW + NBW; 0 picks: enough waves to cover the chip a few times over, else the larger tile (more reuse)
  if (tile == 0) {
    const int64_t rows64 = (M + 63) / 64, rows32 = (M + 31) / 32;
    const int nb = a.nb_total;
    if (rows64 * ((nb + 3) / 4) >= 2048) tile = 24;
    else if (rows64 * ((nb + 1) / 2) >= 2048) tile = 22;
    else if (rows32 * ((nb + 1) / 2) >= 1024) tile = 12;
    else tile = 11;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (tile) {
    case 11: gemm_launch<1, 1>(a, b_layout, st); break;
    case 12: gemm_launch<1, 2>(a, b_layout, st); break;
    case 14: gemm_launch<1, 4>(a, b_layout, st); break;
    case 21: gemm_launch<2, 1>(a, b_layout, st); break;
    case 22: gemm_launch<2, 2>(a, b_layout, st); break;
    case 23: gemm_launch<2, 3>(a, b_layout, st); break;
    case 24: gemm_launch<2, 4>(a, b_layout, st); break;
    default:
      set_error("xeq_gemm_f32: unknown tile %d", tile);
      return XEQ_ERR_INVALID_ARGUMENT;
  }
  XEQ_CHECK_LAUNCH("xeq_gemm_f32");
  return XEQ_OK;
}

}  // extern "C"
