// Dense node-side contractions of the path (scalar_mlp, update_mlp, o3.Linear U|V, dot_lin
// and their reverse products): C[M, N] = A[M, K] x W^T (+ bias, + SiLU) with W stored [N, K], in exact f32 on the
// matrix cores.  Tall-skinny operands: M = nodes (x 2l+1) = 10^4 .. 10^5, K and N in 32 .. 576.
//
// Workgroup tile BM x BN, four waves, K walked in chunks of 32:
//   * global -> LDS with row-contiguous 16-byte loads (8 consecutive threads fetch the 128 bytes a row contributes to
//     a chunk): the first, LDS-free form of this kernel (scratch/xeq_gemm_experiment.hip) loaded the MFMA layout
//     straight from global memory, 32 rows per instruction, and was bound by the texture addresser, not the matrix pipe;
//   * LDS rows are padded to 36 floats; v_mfma_f32_32x32x2_f32 wants A[row i][k], B[k][col i] for lane (i = lane & 31,
//     h = lane >> 5) and k = 2 s + h, but the sum over k is order-free, so half h takes k = 16 h + s of the chunk and a
//     lane's 16 operand values per row block are four ds_read_b128;
//   * the next chunk's global loads are in flight while the current chunk is multiplied; the loop body has no
//     conditional load (the last chunk is peeled), so the waits in front of the MFMAs do not drain the prefetch;
//   * epilogue: bias, optional SiLU with the pre-activation written beside it (removes the activation launch of the
//     reference's Linear -> SiLU -> Linear stacks); the accumulator layout stores 128 contiguous bytes per register.
#include "xeq_common.h"

namespace xeq {

typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef float f32x16g __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* W;     // [N, K] row-major
  const float* bias;
  float* C;
  float* Cpre;
  int64_t M, lda, ldw, ldc;
  int K, N;
  int tiles_n;
  int act;  // 0: none, 1: SiLU (C = silu(z), Cpre = z when given)
};

constexpr int GEMM_KC = 32;
constexpr int GEMM_LD = GEMM_KC + 4;   // padded LDS row (floats)

// BM x BN workgroup tile, waves laid out WR x WC (WR * WC = 4), wave tile (BM / WR) x (BN / WC) in 32 x 32 blocks
template <int BM, int BN, int WR, int WC>
__global__ void __launch_bounds__(256) k_gemm_tn(GemmArgs a) {
  static_assert(WR * WC == 4, "four waves");
  constexpr int RW = BM / WR / 32, CW = BN / WC / 32;
  static_assert(RW >= 1 && CW >= 1, "wave tile of at least one 32 x 32 block");
  constexpr int A4 = BM * (GEMM_KC / 4) / 256, B4 = BN * (GEMM_KC / 4) / 256;   // 16-byte loads per thread and chunk
  static_assert(A4 >= 1 && B4 >= 1, "tile too small for 256 loader threads");
  extern __shared__ float smem[];
  float* As = smem;                                  // [2][BM][GEMM_LD]
  float* Bs = smem + 2 * BM * GEMM_LD;               // [2][BN][GEMM_LD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  const int64_t tile_m = blockIdx.x / a.tiles_n;
  const int tile_n = (int)(blockIdx.x - tile_m * a.tiles_n);
  const int64_t m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // loader role: 16-byte piece c4 of row lr (+ 32 q per further load)
  const int lr = tid >> 3, c4 = tid & 7;
  const float* ga[A4];
  const float* gb[B4];
#pragma unroll
  for (int q = 0; q < A4; ++q) {
    int64_t r = m0 + lr + 32 * q;
    if (r >= a.M) r = a.M - 1;
    ga[q] = a.A + r * a.lda + 4 * c4;
  }
#pragma unroll
  for (int q = 0; q < B4; ++q) {
    int n = n0 + lr + 32 * q;
    if (n >= a.N) n = a.N - 1;
    gb[q] = a.W + (int64_t)n * a.ldw + 4 * c4;
  }
  f32x4g ra[A4], rb[B4];
  auto gload = [&]() {
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      ra[q] = *reinterpret_cast<const f32x4g*>(ga[q]);
      ga[q] += GEMM_KC;
    }
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      rb[q] = *reinterpret_cast<const f32x4g*>(gb[q]);
      gb[q] += GEMM_KC;
    }
  };
  auto sstore = [&](int buf) {
    float* ap = As + buf * (BM * GEMM_LD) + lr * GEMM_LD + 4 * c4;
    float* bp = Bs + buf * (BN * GEMM_LD) + lr * GEMM_LD + 4 * c4;
#pragma unroll
    for (int q = 0; q < A4; ++q) *reinterpret_cast<f32x4g*>(ap + 32 * q * GEMM_LD) = ra[q];
#pragma unroll
    for (int q = 0; q < B4; ++q) *reinterpret_cast<f32x4g*>(bp + 32 * q * GEMM_LD) = rb[q];
  };

  f32x16g acc[RW][CW];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw)
#pragma unroll
    for (int cw = 0; cw < CW; ++cw)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[rw][cw][q] = 0.f;

  auto multiply = [&](int buf) {
    const float* ap = As + buf * (BM * GEMM_LD) + (wr * (BM / WR) + i) * GEMM_LD + 16 * h;
    const float* bp = Bs + buf * (BN * GEMM_LD) + (wc * (BN / WC) + i) * GEMM_LD + 16 * h;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      f32x4g av[RW], bv[CW];
#pragma unroll
      for (int rw = 0; rw < RW; ++rw) av[rw] = *reinterpret_cast<const f32x4g*>(ap + 32 * rw * GEMM_LD + 4 * s4);
#pragma unroll
      for (int cw = 0; cw < CW; ++cw) bv[cw] = *reinterpret_cast<const f32x4g*>(bp + 32 * cw * GEMM_LD + 4 * s4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rw = 0; rw < RW; ++rw)
#pragma unroll
          for (int cw = 0; cw < CW; ++cw)
            acc[rw][cw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[rw][j], bv[cw][j], acc[rw][cw], 0, 0, 0);
    }
  };

  const int nt = a.K / GEMM_KC;
  gload();
  sstore(0);
  __syncthreads();
  int t = 0;
  for (; t + 1 < nt; ++t) {
    gload();                 // chunk t + 1 flies under the MFMAs of chunk t
    multiply(t & 1);
    sstore((t + 1) & 1);
    __syncthreads();
  }
  multiply(t & 1);

  // ---- epilogue: register q of the lane is row (q & 3) + 8 (q >> 2) + 4 h, column i of the 32 x 32 block
#pragma unroll
  for (int cw = 0; cw < CW; ++cw) {
    const int col = n0 + wc * (BN / WC) + 32 * cw + i;
    if (col >= a.N) continue;
    const float bz = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t r = m0 + wr * (BM / WR) + 32 * rw + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (r >= a.M) continue;
        const float z = acc[rw][cw][q] + bz;
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

template <int BM, int BN, int WR, int WC>
static void gemm_launch(GemmArgs a, hipStream_t stream) {
  a.tiles_n = (a.N + BN - 1) / BN;
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  const size_t lds = (size_t)2 * (BM + BN) * GEMM_LD * sizeof(float);
  if (lds > 64 * 1024) {
    static bool raised = false;   // per instantiation
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tn<BM, BN, WR, WC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      raised = true;
    }
  }
  hipLaunchKernelGGL((k_gemm_tn<BM, BN, WR, WC>), dim3((unsigned)(tiles_m * a.tiles_n)), dim3(256), lds, stream, a);
}

extern "C" {

// tile codes: 1 = 128 x 64 (4 x 1 waves), 2 = 64 x 64 (2 x 2), 3 = 128 x 128 (2 x 2), 4 = 64 x 128 (2 x 2), 5 = 32 x 128 (1 x 4),
// 6 = 256 x 32 (4 x 1), 7 = 128 x 32 (4 x 1)
int xeq_gemm_f32(const float* A, int64_t M, int K, int64_t lda, const float* W, int64_t ldw, int N, const float* bias, int act,
                 float* C, int64_t ldc, float* C_pre, int tile, void* stream) {
  XEQ_CHECK_ARG(M >= 0 && K > 0 && N > 0, "xeq_gemm_f32: bad sizes");
  XEQ_CHECK_ARG(K % GEMM_KC == 0, "xeq_gemm_f32: K must be a multiple of 32 (got %d)", K);
  XEQ_CHECK_ARG((lda & 3) == 0 && lda >= K && (ldw & 3) == 0 && ldw >= K && ldc >= N,
                "xeq_gemm_f32: lda / ldw must be multiples of 4 and cover K; ldc must cover N");
  XEQ_CHECK_ARG(act == 0 || act == 1, "xeq_gemm_f32: act is 0 (none) or 1 (SiLU)");
  XEQ_CHECK_ARG((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
                "xeq_gemm_f32: A and W must be 16-byte aligned");
  if (M == 0) return XEQ_OK;
  GemmArgs a{};
  a.A = A; a.W = W; a.bias = bias; a.C = C; a.Cpre = C_pre;
  a.M = M; a.lda = lda; a.ldw = ldw; a.ldc = ldc;
  a.K = K; a.N = N; a.act = act;
  if (tile == 0) tile = N >= 256 ? 1 : 2;
  hipStream_t st = (hipStream_t)stream;
  switch (tile) {
    case 1: gemm_launch<128, 64, 4, 1>(a, st); break;
    case 2: gemm_launch<64, 64, 2, 2>(a, st); break;
    case 3: gemm_launch<128, 128, 2, 2>(a, st); break;
    case 4: gemm_launch<64, 128, 2, 2>(a, st); break;
    case 5: gemm_launch<32, 128, 1, 4>(a, st); break;
    case 6: gemm_launch<256, 32, 4, 1>(a, st); break;
    case 7: gemm_launch<128, 32, 4, 1>(a, st); break;
    default:
      set_error("xeq_gemm_f32: unknown tile %d", tile);
      return XEQ_ERR_INVALID_ARGUMENT;
  }
  XEQ_CHECK_LAUNCH("xeq_gemm_f32");
  return XEQ_OK;
}

}  // extern "C"
