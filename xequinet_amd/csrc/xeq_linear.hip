// Single linear layers of the node side on the matrix cores: the contractions round 2 still gave to library GEMMs.
//   dot_lin                Linear(224, 128, bias=False) on <U, V> and its input gradient   (nn/xpainn.py:191-193, :222-223)
//   embedding              Linear(56, 128) on the gathered table rows                      (nn/xpainn.py:43-48, nn/basic.py:57)
//   energy head            Linear(128, 64) - SiLU - Linear(64, 1) and its input gradient   (nn/output.py:104-118)
// Why own kernels for 0.13 ms of library time: a library picks another GEMM kernel -- another summation order -- for another row
// count, so the same molecule got other bits in another batch (sharded against unsharded, chunked against whole: 4e-4 in the
// forces of an ill-conditioned molecule).  Here a row's sums run in one fixed order whatever the batch: with these entries the
// whole f32 path is batch-independent bit for bit.
//
// Design: the MLP kernels' conventions (xeq_mlp.hip).  A workgroup (4 waves) owns 32 consecutive rows; exact-f32
// v_mfma_f32_32x32x2_f32 tiles D[column][row] with the WEIGHT fragment as the A operand, read global -> register from the copy
// xeq_mlp_pack makes (packed[tile][k-group][lane][4], the bias as one more k-group against a row of ones); the row operand is
// staged once in LDS (K <= 256) with 16-byte loads, optionally gathered through a row index (the embedding table lookup).
// Wave w computes output tiles w, w + 4.  These are small products (K <= 224, at most 8 output tiles): one k-chain per tile.
#include "xeq_common.h"
#include "xeq_linear_s.h"

namespace xeq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LIN_ROWS = 32;


__global__ void __launch_bounds__(256) k_linear(LinArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[LIN_ROWS * LIN_XLD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * LIN_ROWS;
  const int rows_here = (int)min((int64_t)LIN_ROWS, a.n - row0);
  const int k4 = a.K >> 2;   // float4 per row
  for (int idx = tid; idx < LIN_ROWS * k4; idx += 256) {
    const int r = idx / k4, c4 = idx - r * k4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows_here) {
      int64_t src = row0 + r;
      if (a.row_index) src = a.row_index[src];
      v = *reinterpret_cast<const float4*>(a.X + src * a.ldx + 4 * c4);
    }
    *reinterpret_cast<float4*>(&Xs[r * LIN_XLD + 4 * c4]) = v;
  }
  __syncthreads();
  const int G = a.K >> 3, nt = a.n_out >> 5;
  const float one_k0 = kh == 0 ? 1.f : 0.f;
  const float* xs = &Xs[i * LIN_XLD + 4 * kh];
  const bool row_ok = i < rows_here;
  for (int t = wave; t < nt; t += 4) {
    const float4* wp = reinterpret_cast<const float4*>(a.Wp) + (int64_t)t * (G + 1) * 64 + lane;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // weight fragments four k-groups ahead of their MFMAs (an L2 round trip is ~500 cycles, a group's four MFMAs 256)
    float4 w0 = wp[0], w1 = wp[(1 < G ? 1 : G - 1) * 64], w2 = wp[(2 < G ? 2 : G - 1) * 64], w3 = wp[(3 < G ? 3 : G - 1) * 64];
    for (int q = 0; q < G; ++q) {
      const float4 wn = wp[(q + 4 < G ? q + 4 : G - 1) * 64];
      const float4 xv = *reinterpret_cast<const float4*>(xs + 8 * q);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, xv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, xv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, xv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, xv.w, acc, 0, 0, 0);
      w0 = w1;
      w1 = w2;
      w2 = w3;
      w3 = wn;
    }
    if (a.has_bias) {
      const float bias_a = reinterpret_cast<const float*>(wp + (int64_t)G * 64)[0];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, one_k0, acc, 0, 0, 0);
    }
    if (row_ok) {
      const int64_t row = row0 + i;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 32 * t + 8 * g + 4 * kh;
        float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        if (a.pre) *reinterpret_cast<float4*>(a.pre + row * a.n_out + col) = v;
        if (a.act == 1) v = make_float4(lin_silu(v.x), lin_silu(v.y), lin_silu(v.z), lin_silu(v.w));
        *reinterpret_cast<float4*>(a.Y + row * a.ldy + col) = v;
      }
    }
  }
}

// (the few-row form k_linear_s: xeq_linear_s.h)
__global__ void __launch_bounds__(256) k_linear_s(LinArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[LIN_S_ROWS * LIN_XLD];
  linear_s_body<4>(a, Xs, (int)blockIdx.x, (int)threadIdx.x);
}

// atomic energies of the head's last layer: out[n] = <hidden[n, :], w2> + b2 (+ add[n]); one 16-lane group per node, H <= 64 x 16
__global__ void k_head_dot(const float* __restrict__ hidden, int64_t n, int H, const float* __restrict__ w2, const float* __restrict__ b2,
                           float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t node = t >> 4;
  const int sub = (int)(t & 15);
  float acc = 0.f;
  if (node < n)
    for (int c = 4 * sub; c < H; c += 64) {   // fixed order per node: lane `sub` takes columns 4 sub + 64 k .. + 3
      const float4 h = *reinterpret_cast<const float4*>(hidden + node * H + c);
      const float4 w = *reinterpret_cast<const float4*>(w2 + c);
      acc = fmaf(h.x, w.x, acc);
      acc = fmaf(h.y, w.y, acc);
      acc = fmaf(h.z, w.z, acc);
      acc = fmaf(h.w, w.w, acc);
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (node < n && sub == 0) out[node] = acc + (b2 ? b2[0] : 0.f);
}

// reverse of the head's last two stages: g_hidden[n, j] = g_atomic[n] w2[j] silu'(pre[n, j])   (g_atomic NULL: ones, dE/d atomic_i = 1)
__global__ void k_head_bwd_hidden(const float* __restrict__ pre, int64_t n, int H, const float* __restrict__ w2,
                                  const float* __restrict__ g_atomic, float* __restrict__ g_hidden) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4
  const int h4 = H >> 2;
  if (t >= n * h4) return;
  const int64_t node = t / h4;
  const int c = 4 * (int)(t - node * h4);
  const float ga = g_atomic ? g_atomic[node] : 1.f;
  const float4 p = *reinterpret_cast<const float4*>(pre + node * H + c);
  const float4 w = *reinterpret_cast<const float4*>(w2 + c);
  auto dsilu = [](float x) {   // aten silu_backward: sig (1 + x (1 - sig))
    const float sig = 1.f / (1.f + expf(-x));
    return sig * (1.f + x * (1.f - sig));
  };
  *reinterpret_cast<float4*>(g_hidden + node * H + c) =
      make_float4(ga * w.x * dsilu(p.x), ga * w.y * dsilu(p.y), ga * w.z * dsilu(p.z), ga * w.w * dsilu(p.w));
}

// ---- the whole energy head of a force evaluation in one launch (round 5) -------------------------------------------------------------
// EnergyOut's MLP (nn/output.py:104-118) on 32 rows per workgroup: pre = W1 s + b1 (exact-f32 tiles as k_linear), hidden = SiLU(pre),
// atomic = <hidden, w2> + b2, and -- because the head's output is ONE number per node -- its whole reverse pass as a saved row
// J[n, :] = d atomic_n / d s_n = W1^T (w2 . SiLU'(pre)): the reverse pass of a force evaluation (nn/basic.py:143-159) is then
// g_s[n, :] = g[n] J[n, :] (k_head_bwd).  Replaces k_linear + k_head_dot (forward) and k_head_bwd_hidden + k_linear (reverse).
// A row's sums run in one fixed order whatever the batch.
struct HeadArgs {
  const float* S;       // [n, lds]
  int64_t lds, n;
  int F, H;             // node_dim, hidden width (multiples of 32, <= 256)
  const float* W1p;     // xeq_mlp_pack(W1 [H, F], b1): [H / 32][F / 8 + 1][64][4]
  const float* W1tp;    // xeq_mlp_pack(W1, transposed): the product g_hidden [., H] x W1 [H, F]: [F / 32][H / 8 + 1][64][4] (no bias)
  const float* w2;      // [H]
  const float* b2;      // [1] or NULL
  float* atomic;        // [n]
  float* J;             // [n, F] or NULL (energies only)
};

__global__ void __launch_bounds__(256) k_head_fused(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float head_lds[];
  const int XLD = a.F + 4, HLD = a.H + 4;
  float* Xs = head_lds;                   // [32][F + 4] staged rows
  float* Es = Xs + LIN_ROWS * XLD;        // [32][H + 4] hidden . w2
  float* Gs = Es + LIN_ROWS * HLD;        // [32][H + 4] w2 . SiLU'(pre)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * LIN_ROWS;
  const int rows_here = (int)min((int64_t)LIN_ROWS, a.n - row0);
  const int k4 = a.F >> 2;
  for (int idx = tid; idx < LIN_ROWS * k4; idx += 256) {
    const int r = idx / k4, c4 = idx - r * k4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows_here) v = *reinterpret_cast<const float4*>(a.S + (row0 + r) * a.lds + 4 * c4);
    *reinterpret_cast<float4*>(&Xs[r * XLD + 4 * c4]) = v;
  }
  __syncthreads();
  const float one_k0 = kh == 0 ? 1.f : 0.f;
  {  // hidden layer: a wave takes (output tile t of 32 hidden columns, half kp of the k range) when the tiles leave waves idle (the
     // default head: two tiles, four waves), so the serial chain of a wave is half as long; the two partial sums meet in LDS, in one order
    const int G = a.F >> 3, nt = a.H >> 5, ksplit = (nt <= 2 && (G & 1) == 0) ? 2 : 1;
    const float* xs = &Xs[i * XLD + 4 * kh];
    for (int item = wave; item < nt * ksplit; item += 4) {
      const int t = item % nt, kp = item / nt;
      const int q0 = kp * (G / ksplit), q1 = q0 + G / ksplit;
      const float4* wp = reinterpret_cast<const float4*>(a.W1p) + (int64_t)t * (G + 1) * 64 + lane;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      auto wq = [&](int q) { return wp[(q < q1 ? q : q1 - 1) * 64]; };
      float4 w0 = wq(q0), w1 = wq(q0 + 1), w2r = wq(q0 + 2), w3 = wq(q0 + 3);
      for (int q = q0; q < q1; ++q) {
        const float4 wn = wq(q + 4);
        const float4 xv = *reinterpret_cast<const float4*>(xs + 8 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, xv.w, acc, 0, 0, 0);
        w0 = w1;
        w1 = w2r;
        w2r = w3;
        w3 = wn;
      }
      if (kp == 0) {   // the bias rides with the first half
        const float bias_a = reinterpret_cast<const float*>(wp + (int64_t)G * 64)[0];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, one_k0, acc, 0, 0, 0);
      }
      float* dst = kp == 0 ? Es : Gs;   // partial pre-activations
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(&dst[i * HLD + 32 * t + 8 * g + 4 * kh]) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();
    // pre = half 0 (+ half 1); e = SiLU(pre) w2 -> Es, w2 SiLU'(pre) -> Gs, in place: a thread per four columns
    const int h4 = a.H >> 2;
    for (int idx = tid; idx < LIN_ROWS * h4; idx += 256) {
      const int r = idx / h4, col = 4 * (idx - r * h4);
      float4 p = *reinterpret_cast<const float4*>(&Es[r * HLD + col]);
      if (ksplit == 2) {
        const float4 p1 = *reinterpret_cast<const float4*>(&Gs[r * HLD + col]);
        p = make_float4(p.x + p1.x, p.y + p1.y, p.z + p1.z, p.w + p1.w);
      }
      const float4 wv = *reinterpret_cast<const float4*>(a.w2 + col);
      const float xq[4] = {p.x, p.y, p.z, p.w}, wq4[4] = {wv.x, wv.y, wv.z, wv.w};
      float e[4], gh[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x = xq[c];
        const float sig = 1.f / (1.f + expf(-x));
        e[c] = (x * sig) * wq4[c];                               // SiLU(pre) w2
        gh[c] = wq4[c] * (sig * (1.f + x * (1.f - sig)));        // w2 SiLU'(pre)  (aten silu_backward's form)
      }
      *reinterpret_cast<float4*>(&Es[r * HLD + col]) = make_float4(e[0], e[1], e[2], e[3]);
      *reinterpret_cast<float4*>(&Gs[r * HLD + col]) = make_float4(gh[0], gh[1], gh[2], gh[3]);
    }
  }
  __syncthreads();
  {  // atomic energies: eight threads per row, columns sub, sub + 8, ... then a butterfly: one fixed order per row
    const int r = tid >> 3, sub = tid & 7;
    float acc = 0.f;
    for (int c = sub; c < a.H; c += 8) acc += Es[r * HLD + c];
    acc += __shfl_xor(acc, 4, 8);
    acc += __shfl_xor(acc, 2, 8);
    acc += __shfl_xor(acc, 1, 8);
    if (sub == 0 && r < rows_here) a.atomic[row0 + r] = acc + (a.b2 ? a.b2[0] : 0.f);
  }
  if (a.J) {  // J = (w2 . SiLU'(pre)) W1: output tile t (32 input columns) per wave, K = H
    const int G = a.H >> 3, nt = a.F >> 5;
    const float* gs = &Gs[i * HLD + 4 * kh];
    for (int t = wave; t < nt; t += 4) {
      const float4* wp = reinterpret_cast<const float4*>(a.W1tp) + (int64_t)t * (G + 1) * 64 + lane;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float4 w0 = wp[0], w1 = wp[(1 < G ? 1 : G - 1) * 64], w2r = wp[(2 < G ? 2 : G - 1) * 64], w3 = wp[(3 < G ? 3 : G - 1) * 64];
      for (int q = 0; q < G; ++q) {
        const float4 wn = wp[(q + 4 < G ? q + 4 : G - 1) * 64];
        const float4 xv = *reinterpret_cast<const float4*>(gs + 8 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, xv.w, acc, 0, 0, 0);
        w0 = w1;
        w1 = w2r;
        w2r = w3;
        w3 = wn;
      }
      if (i < rows_here) {
        const int64_t row = row0 + i;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(a.J + row * a.F + 32 * t + 8 * g + 4 * kh) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      }
    }
  }
}

// reverse of the head for whatever arrives at its two outputs: g_s[n, :] = (g_atomic[n] + g_total[graph(n)]) J[n, :]  (either may be NULL;
// strides in elements: 0 = one broadcast value, what autograd hands over for the sum of the energies)
__global__ void k_head_bwd(const float* __restrict__ J, int64_t n, int F, const float* __restrict__ g_atomic, int64_t ga_stride,
                           const float* __restrict__ g_total, int64_t gt_stride, const int64_t* __restrict__ batch, float* __restrict__ g_s) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4
  const int f4 = F >> 2;
  if (t >= n * f4) return;
  const int64_t node = t / f4;
  float g = g_atomic ? g_atomic[node * ga_stride] : 0.f;
  if (g_total) g += g_total[(batch ? batch[node] : 0) * gt_stride];
  const float4 j = reinterpret_cast<const float4*>(J)[t];
  reinterpret_cast<float4*>(g_s)[t] = make_float4(g * j.x, g * j.y, g * j.z, g * j.w);
}

// ---- weight gradients of a training pass: dW[M, K] = A[n, M]^T B[n, K] over the n node rows ----------------------------------------
// (A = dL/dy rows, B = the layer's input rows: nn.Linear's weight gradient; the o3.Linear blocks likewise on the BT views.)  The
// reduction runs over the LONG dimension (n = 18 k rows against M, K <= 576; the library takes 60-130 us per product here,
// scratch/bench_wgrad.py, and another summation order for another row count); here the rows are cut into chunks, a wave owns a 64 x 64 block of dW for one chunk -- four exact-f32 32x32x2 tiles, each MFMA taking
// two rows, the operands read straight from global memory (consecutive waves of a workgroup share the chunk: L1 / L2 hits) -- and
// writes parts[chunk][M][K]; the caller sums the parts in chunk order: fixed summation order, bitwise reproducible.
struct WgradArgs {
  const float* A;
  const float* B;
  int64_t lda, ldb, n, rows_per_chunk;
  int M, K, bm, bk, n_chunks;
  int with_bias;     // also the column sums of A (the layer's bias gradient): M more floats behind each chunk's [M, K] block
  float* parts;
};

__global__ void __launch_bounds__(256) k_wgrad(WgradArgs a) {
  const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nblocks = a.bm * a.bk;
  const int64_t chunk = wave / nblocks;
  if (chunk >= a.n_chunks) return;
  const int blk = (int)(wave - chunk * nblocks);
  const int m0 = 64 * (blk / a.bk), k0 = 64 * (blk % a.bk);
  const int64_t r0 = chunk * a.rows_per_chunk, r1 = min(a.n, r0 + a.rows_per_chunk);
  const bool am0 = m0 + i < a.M, am1 = m0 + 32 + i < a.M, bk0 = k0 + i < a.K, bk1 = k0 + 32 + i < a.K;
  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
  const float* pa = a.A + m0 + i;
  const float* pb = a.B + k0 + i;
  constexpr int UN = 4;   // row pairs in flight
  float sa0 = 0.f, sa1 = 0.f;   // column sums of A over this lane's rows (bias gradient; used by the k0 = 0 blocks)
  for (int64_t r = r0; r < r1; r += 2 * UN) {
    float va0[UN], va1[UN], vb0[UN], vb1[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t row = r + 2 * u + kh;
      const bool ok = row < r1;
      va0[u] = (ok && am0) ? pa[row * a.lda] : 0.f;
      va1[u] = (ok && am1) ? pa[row * a.lda + 32] : 0.f;
      vb0[u] = (ok && bk0) ? pb[row * a.ldb] : 0.f;
      vb1[u] = (ok && bk1) ? pb[row * a.ldb + 32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      sa0 += va0[u];
      sa1 += va1[u];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0[u], vb0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0[u], vb1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1[u], vb0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1[u], vb1[u], acc[1][1], 0, 0, 0);
    }
  }
  const int64_t stride = (int64_t)a.M * a.K + (a.with_bias ? a.M : 0);
  float* out = a.parts + chunk * stride;
  if (a.with_bias && k0 == 0) {   // one column block per row block writes the sums: odd + even rows of the chunk
    sa0 += __shfl_xor(sa0, 32, 64);
    sa1 += __shfl_xor(sa1, 32, 64);
    if (kh == 0) {
      if (am0) out[(int64_t)a.M * a.K + m0 + i] = sa0;
      if (am1) out[(int64_t)a.M * a.K + m0 + 32 + i] = sa1;
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int col = k0 + 32 * v + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowm = m0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (rowm < a.M && col < a.K) out[(int64_t)rowm * a.K + col] = acc[u][v][r];
      }
    }
}

// ---- the hardware property the few-row forms rest on, as a test entry (xeq_mfma_order_probe) -------------------------------------------
// D = A B over K = 224 for one 32 x 32 output block: once with v_mfma_f32_32x32x2_f32 (the 32-row kernels' instruction), once with
// v_mfma_f32_16x16x4_f32 on its four 16 x 16 quarters (the few-row kernels'), once as a sequential fmaf chain per output element on the
// vector pipe; counts the outputs whose BITS differ.
constexpr int PROBE_K = 224;
__global__ void __launch_bounds__(256) k_mfma_order_probe(const float* __restrict__ A, const float* __restrict__ B, int32_t* __restrict__ diff) {
  __shared__ float d32[1024], d16[1024];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    const int i = l & 31, kh = l >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < PROBE_K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * PROBE_K + k + kh], B[(k + kh) * 32 + i], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) d32[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = acc[r];
  }
  {
    const int i = l & 15, kq = l >> 4, bi = wave >> 1, bj = wave & 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < PROBE_K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * bi + i) * PROBE_K + k + kq], B[(k + kq) * 32 + 16 * bj + i], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) d16[(16 * bi + 4 * kq + r) * 32 + 16 * bj + i] = acc[r];
  }
  __syncthreads();
  int n_shape = 0, n_chain = 0;
  for (int x = tid; x < 1024; x += 256) {
    const int i = x >> 5, j = x & 31;
    float s = 0.f;
    for (int k = 0; k < PROBE_K; ++k) s = __builtin_fmaf(A[i * PROBE_K + k], B[k * 32 + j], s);
    n_shape += __float_as_uint(d32[x]) != __float_as_uint(d16[x]);
    n_chain += __float_as_uint(d32[x]) != __float_as_uint(s);
  }
  if (n_shape) atomicAdd(&diff[0], n_shape);
  if (n_chain) atomicAdd(&diff[1], n_chain);
}

}  // namespace xeq

using namespace xeq;

extern "C" {

/* test entry: a [32, 224] and b [224, 32] f32 on the device; diff[0] += outputs of a b whose bits differ between v_mfma_f32_32x32x2_f32 and
 * v_mfma_f32_16x16x4_f32, diff[1] += outputs that differ between the former and a sequential fmaf chain (diff: two int32, zeroed by the caller) */
int xeq_mfma_order_probe(const float* a, const float* b, int32_t* diff, void* stream) {
  XEQ_CHECK_ARG(a && b && diff, "xeq_mfma_order_probe: null buffer");
  hipLaunchKernelGGL(k_mfma_order_probe, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, diff);
  XEQ_CHECK_LAUNCH("xeq_mfma_order_probe");
  return XEQ_OK;
}

int xeq_linear_supported(int dtype, int k_in, int n_out) {
  return dtype == XEQ_F32 && k_in >= 8 && k_in % 8 == 0 && k_in <= LIN_KMAX && n_out >= 32 && n_out % 32 == 0 && n_out <= 256 ? 1 : 0;
}

int xeq_linear_fwd(const void* x, int64_t ldx, int64_t n, int k_in, const int32_t* row_index, const void* w_packed, int n_out,
                   int has_bias, int act, void* pre, void* y, int64_t ldy, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && xeq_linear_supported(XEQ_F32, k_in, n_out), "xeq_linear_fwd: K = %d (multiple of 8, <= %d), n_out = %d (multiple of 32, <= 256)",
                k_in, LIN_KMAX, n_out);
  XEQ_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= k_in && ldy >= n_out && (act == 0 || act == 1), "xeq_linear_fwd: row strides must be multiples of four floats");
  if (n == 0) return XEQ_OK;
  LinArgs a{(const float*)x, row_index, ldx, n, k_in, n_out, (const float*)w_packed, has_bias, act, (float*)pre, (float*)y, ldy};
  if (n <= xeq_small_rows())
    hipLaunchKernelGGL(k_linear_s, dim3((unsigned)((n + LIN_S_ROWS - 1) / LIN_S_ROWS) * (unsigned)((n_out + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_linear, dim3((unsigned)((n + LIN_ROWS - 1) / LIN_ROWS)), dim3(256), 0, (hipStream_t)stream, a);
  XEQ_CHECK_LAUNCH("xeq_linear_fwd");
  return XEQ_OK;
}

int xeq_head_dot(const void* hidden, int64_t n, int hidden_dim, const void* w2, const void* b2, void* out, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && hidden_dim >= 4 && hidden_dim % 4 == 0, "xeq_head_dot: hidden width %d", hidden_dim);
  if (n == 0) return XEQ_OK;
  hipLaunchKernelGGL(k_head_dot, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)hidden, n,
                     hidden_dim, (const float*)w2, (const float*)b2, (float*)out);
  XEQ_CHECK_LAUNCH("xeq_head_dot");
  return XEQ_OK;
}

int xeq_head_bwd_hidden(const void* pre, int64_t n, int hidden_dim, const void* w2, const void* g_atomic, void* g_hidden, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && hidden_dim >= 4 && hidden_dim % 4 == 0, "xeq_head_bwd_hidden: hidden width %d", hidden_dim);
  if (n == 0) return XEQ_OK;
  const int64_t total = n * (hidden_dim / 4);
  hipLaunchKernelGGL(k_head_bwd_hidden, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)pre, n,
                     hidden_dim, (const float*)w2, (const float*)g_atomic, (float*)g_hidden);
  XEQ_CHECK_LAUNCH("xeq_head_bwd_hidden");
  return XEQ_OK;
}

int xeq_head_supported(int dtype, int node_dim, int hidden_dim) {
  return dtype == XEQ_F32 && xeq_linear_supported(dtype, node_dim, hidden_dim) && xeq_linear_supported(dtype, hidden_dim, node_dim) ? 1 : 0;
}

int xeq_head_fwd(const void* s, int64_t lds, int64_t n, int node_dim, int hidden_dim, const void* w1_packed, const void* w1t_packed,
                 const void* w2, const void* b2, void* atomic, void* jac, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && xeq_head_supported(XEQ_F32, node_dim, hidden_dim), "xeq_head_fwd: node_dim %d, hidden %d (multiples of 32, <= 256)", node_dim, hidden_dim);
  XEQ_CHECK_ARG(s && w1_packed && w2 && atomic && (w1t_packed || !jac) && lds >= node_dim && lds % 4 == 0, "xeq_head_fwd: bad arguments");
  if (n == 0) return XEQ_OK;
  HeadArgs a{(const float*)s, lds, n, node_dim, hidden_dim, (const float*)w1_packed, (const float*)w1t_packed, (const float*)w2, (const float*)b2,
             (float*)atomic, (float*)jac};
  const size_t shmem = sizeof(float) * (size_t)LIN_ROWS * ((node_dim + 4) + 2 * (hidden_dim + 4));
  hipLaunchKernelGGL(k_head_fused, dim3((unsigned)((n + LIN_ROWS - 1) / LIN_ROWS)), dim3(256), shmem, (hipStream_t)stream, a);
  XEQ_CHECK_LAUNCH("xeq_head_fwd");
  return XEQ_OK;
}

int xeq_head_bwd(const void* jac, int64_t n, int node_dim, const void* g_atomic, int64_t ga_stride, const void* g_total, int64_t gt_stride,
                 const int64_t* batch, void* g_s, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && node_dim >= 4 && node_dim % 4 == 0 && jac && g_s && (g_atomic || g_total), "xeq_head_bwd: bad arguments");
  if (n == 0) return XEQ_OK;
  const int64_t total = n * (node_dim / 4);
  hipLaunchKernelGGL(k_head_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)jac, n, node_dim,
                     (const float*)g_atomic, ga_stride, (const float*)g_total, gt_stride, batch, (float*)g_s);
  XEQ_CHECK_LAUNCH("xeq_head_bwd");
  return XEQ_OK;
}

/* chunks xeq_wgrad cuts n rows into for an [M, K] gradient: enough (block, chunk) waves for ~2 per SIMD, at least 128 rows each */
int xeq_wgrad_chunks(int64_t n, int m, int k) {
  const int64_t blocks = (int64_t)((m + 63) / 64) * ((k + 63) / 64);
  int64_t c = (2048 + blocks - 1) / blocks, cmax = (n + 127) / 128;
  if (c > cmax) c = cmax;
  return (int)(c < 1 ? 1 : c);
}

int xeq_wgrad(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t n, int m, int k, int with_bias, int n_chunks, void* parts,
              void* stream) {
  XEQ_CHECK_ARG(n >= 0 && m >= 1 && k >= 1 && lda >= m && ldb >= k, "xeq_wgrad: bad shape n = %lld, M = %d, K = %d", (long long)n, m, k);
  XEQ_CHECK_ARG(n_chunks == xeq_wgrad_chunks(n, m, k), "xeq_wgrad: parts must hold xeq_wgrad_chunks(n, M, K) = %d blocks, got %d",
                xeq_wgrad_chunks(n, m, k), n_chunks);
  WgradArgs w{(const float*)a, (const float*)b, lda, ldb, n, (n + n_chunks - 1) / n_chunks, m, k, (m + 63) / 64, (k + 63) / 64, n_chunks,
              with_bias ? 1 : 0, (float*)parts};
  if (w.rows_per_chunk & 1) ++w.rows_per_chunk;   // an MFMA takes two rows: chunks start on even rows
  const int64_t waves = (int64_t)w.bm * w.bk * n_chunks;
  hipLaunchKernelGGL(k_wgrad, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w);
  XEQ_CHECK_LAUNCH("xeq_wgrad");
  return XEQ_OK;
}

}  // extern "C"
