// General Clebsch-Gordan tensor product, one instruction (path) per launch  --  the e3nn o3.TensorProduct the reference
// builds from get_feasible_tp (nn/tp.py:20-107) for SelfMixTP (nn/xe3net.py:133-146, 'uuu', internal weights) and the
// Cartesian-tensor head (nn/output.py:411-421, 'uuw', per-sample weights).  SURVEY 8f-3.
//
//   out[n, w, k] += coeff * sum_{u, v} W[...] * sum_{i, j} C[i, j, k] x1[n, u, i] x2[n, v, j]
//
// with the index pattern of the connection mode (u, v, w -> which multiplicities are tied) and C the real Wigner-3j table
// of (l1, l2, l3) (xequinet_amd/data/wigner3j_lmax4.npz).  A thread owns one output element (n, w, k); launches of the
// paths of one product run in stream order and accumulate into `out`, so the sum over paths has a fixed order
// (deterministic, no atomics).  The contraction is small and ragged (2l+1 <= 9): this op is HBM / latency bound, not a GEMM.
#include <stdlib.h>

#include "xeq_common.h"

namespace xeq {

enum { TP_UVW = 0, TP_UVU = 1, TP_UVV = 2, TP_UUW = 3, TP_UUU = 4, TP_UVUV = 5 };

struct TpPath {
  int64_t n;
  int dim1, dim2, dim_out;      // row strides of x1, x2, out
  int off1, off2, offo;         // first element of the instruction's irrep blocks
  int mul1, mul2, mulo;
  int d1, d2, d3;               // 2 l + 1
  int mode, has_weight;
  int64_t w_stride;             // 0: shared weights; otherwise floats per sample
};

template <typename T>
__global__ void k_tp_path(TpPath p, const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ cg,
                          const T* __restrict__ weight, T coeff, T* __restrict__ out) {
  const int64_t per_node = (int64_t)p.mulo * p.d3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.n * per_node) return;
  const int64_t n = t / per_node;
  const int r = (int)(t - n * per_node), w = r / p.d3, k = r - w * p.d3;
  const T* a = x1 + n * p.dim1 + p.off1;
  const T* b = x2 + n * p.dim2 + p.off2;
  const T* W = p.has_weight ? weight + n * p.w_stride : nullptr;
  // bilinear form of the 3j slice k between row u of x1 and row v of x2
  auto pair = [&](int u, int v) {
    T s = T(0);
    for (int i = 0; i < p.d1; ++i) {
      const T ai = a[u * p.d1 + i];
      T q = T(0);
      for (int j = 0; j < p.d2; ++j) q += cg[(i * p.d2 + j) * p.d3 + k] * b[v * p.d2 + j];
      s += ai * q;
    }
    return s;
  };
  T acc = T(0);
  switch (p.mode) {
    case TP_UUU:
      acc = (W ? W[w] : T(1)) * pair(w, w);
      break;
    case TP_UUW:
      for (int u = 0; u < p.mul1; ++u) acc += (W ? W[u * p.mulo + w] : T(1)) * pair(u, u);
      break;
    case TP_UVW:
      for (int u = 0; u < p.mul1; ++u)
        for (int v = 0; v < p.mul2; ++v) acc += (W ? W[(u * p.mul2 + v) * p.mulo + w] : T(1)) * pair(u, v);
      break;
    case TP_UVU:
      for (int v = 0; v < p.mul2; ++v) acc += (W ? W[w * p.mul2 + v] : T(1)) * pair(w, v);
      break;
    case TP_UVV:
      for (int u = 0; u < p.mul1; ++u) acc += (W ? W[u * p.mul2 + w] : T(1)) * pair(u, w);
      break;
    default: {   // TP_UVUV: w = u * mul2 + v
      const int u = w / p.mul2, v = w - u * p.mul2;
      acc = (W ? W[w] : T(1)) * pair(u, v);
    }
  }
  out[n * p.dim_out + p.offo + r] += coeff * acc;
}


// ------------------------------------------------------------------------------------------------ all paths of a product in ONE launch
// (round 4).  A workgroup owns a tile of TN consecutive nodes: their x1 and x2 rows (contiguous in memory) and every
// 3j table of the product are staged in LDS once; a thread then owns an output element (node, w, k) -- or, for the modes that sum
// over the input multiplicities into every w ('uvw', 'uuw'), a block of WB consecutive w of one (node, k), so that the bilinear
// form pair(u, v, k) is formed once per block instead of once per w -- walks ALL paths that end in that output block in their
// instruction order (the order the one-launch-per-path form accumulated in) and stores the sum once: no read-modify-write of
// `out`, no launch per path, inputs read from LDS.
constexpr int TP_MAXP = 40, TP_MAXS = 16, TP_WB = 4;
struct TpPathF {
  int off1, off2;
  int mul1, mul2;
  int d1, d2;
  int mode, cg_off;      // first float of the path's 3j table in the staged tables
  int w_off;             // first weight of the path (floats), -1: unweighted
  double coeff;
};
struct TpSlot {
  int offo, mulo, d3, blocked;   // blocked: a thread owns TP_WB consecutive w
  int p0, p1;                    // paths [p0, p1)
};
struct TpArgsF {
  int64_t n;
  int dim1, dim2, dim_out, tn;
  int n_slots, cg_floats;
  int64_t w_stride;
  TpSlot slot[TP_MAXS];
  TpPathF path[TP_MAXP];
};

template <typename T>
__global__ void __launch_bounds__(256) k_tp_fused(TpArgsF a, const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ cg,
                                                  const T* __restrict__ weight, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tp_lds[];
  T* s1 = reinterpret_cast<T*>(tp_lds);
  T* s2 = s1 + (int64_t)a.tn * a.dim1;
  T* sc = s2 + (int64_t)a.tn * a.dim2;
  const int64_t n0 = (int64_t)blockIdx.x * a.tn;
  const int tn = (int)min((int64_t)a.tn, a.n - n0);
  for (int i = threadIdx.x; i < tn * a.dim1; i += blockDim.x) s1[i] = x1[n0 * a.dim1 + i];
  for (int i = threadIdx.x; i < tn * a.dim2; i += blockDim.x) s2[i] = x2[n0 * a.dim2 + i];
  for (int i = threadIdx.x; i < a.cg_floats; i += blockDim.x) sc[i] = cg[i];
  __syncthreads();
  for (int si = 0; si < a.n_slots; ++si) {
    const TpSlot sl = a.slot[si];
    const int wblocks = sl.blocked ? (sl.mulo + TP_WB - 1) / TP_WB : sl.mulo;
    const int per_node = wblocks * sl.d3;
    for (int e = threadIdx.x; e < tn * per_node; e += blockDim.x) {
      const int nl = e / per_node, r = e - nl * per_node, wb = r / sl.d3, k = r - wb * sl.d3;
      const int w0 = sl.blocked ? wb * TP_WB : wb;
      const T* A = s1 + nl * a.dim1;
      const T* B = s2 + nl * a.dim2;
      T tot[TP_WB];
#pragma unroll
      for (int q = 0; q < TP_WB; ++q) tot[q] = T(0);
      for (int pi = sl.p0; pi < sl.p1; ++pi) {
        const TpPathF p = a.path[pi];
        const T* Ap = A + p.off1;
        const T* Bp = B + p.off2;
        const T* C = sc + p.cg_off;
        const T* W = p.w_off >= 0 ? weight + (n0 + nl) * a.w_stride + p.w_off : nullptr;
        auto pair = [&](int u, int v) {
          T s = T(0);
          for (int i = 0; i < p.d1; ++i) {
            const T ai = Ap[u * p.d1 + i];
            T q = T(0);
            for (int j = 0; j < p.d2; ++j) q += C[(i * p.d2 + j) * sl.d3 + k] * Bp[v * p.d2 + j];
            s += ai * q;
          }
          return s;
        };
        T acc[TP_WB];
#pragma unroll
        for (int q = 0; q < TP_WB; ++q) acc[q] = T(0);
        switch (p.mode) {
          case TP_UUU:
            acc[0] = (W ? W[w0] : T(1)) * pair(w0, w0);
            break;
          case TP_UVU:
            for (int v = 0; v < p.mul2; ++v) acc[0] += (W ? W[w0 * p.mul2 + v] : T(1)) * pair(w0, v);
            break;
          case TP_UVV:
            for (int u = 0; u < p.mul1; ++u) acc[0] += (W ? W[u * p.mul2 + w0] : T(1)) * pair(u, w0);
            break;
          case TP_UUW:
            for (int u = 0; u < p.mul1; ++u) {
              const T z = pair(u, u);
#pragma unroll
              for (int q = 0; q < TP_WB; ++q)
                if (w0 + q < sl.mulo) acc[q] += (W ? W[u * sl.mulo + w0 + q] : T(1)) * z;
            }
            break;
          case TP_UVW:
            for (int u = 0; u < p.mul1; ++u)
              for (int v = 0; v < p.mul2; ++v) {
                const T z = pair(u, v);
#pragma unroll
                for (int q = 0; q < TP_WB; ++q)
                  if (w0 + q < sl.mulo) acc[q] += (W ? W[(u * p.mul2 + v) * sl.mulo + w0 + q] : T(1)) * z;
              }
            break;
          default: {   // TP_UVUV: w = u * mul2 + v
            const int u = w0 / p.mul2, v = w0 - u * p.mul2;
            acc[0] = (W ? W[w0] : T(1)) * pair(u, v);
          }
        }
#pragma unroll
        for (int q = 0; q < TP_WB; ++q) tot[q] += (T)p.coeff * acc[q];
      }
      T* o = out + (n0 + nl) * a.dim_out + sl.offo;
      if (sl.blocked) {
#pragma unroll
        for (int q = 0; q < TP_WB; ++q)
          if (w0 + q < sl.mulo) o[(w0 + q) * sl.d3 + k] = tot[q];
      } else {
        o[w0 * sl.d3 + k] = tot[0];
      }
    }
  }
}

// The same product for the modes whose multiplicity index u is TIED between an input and the output or summed into a few outputs --
// 'uuu', 'uuw' (what the reference builds: nn/xe3net.py:133-146, nn/output.py:411-421) and the transposed contractions of their reverse
// passes ('uvu', 'uvv') -- with a THREAD PER (node, u): a wave owns a node, a lane a channel u.  The lane stages the 2 l + 1 values of
// its channel of x1 and x2 in LDS once per path, forms ALL 2 l3 + 1 outputs of the path from them (every product a_i b_j is used for
// all k; the 3j entries are wave-uniform: scalar loads), keeps the sums of the paths into one output block in registers and stores
// them once.  'uuw' sums over u: the lanes' partial sums are added across the wave in a fixed butterfly order (deterministic).
// Against the element-per-thread form each a_i, b_j is read once instead of once per (w, k), and a node's 2 l + 1 runs are contiguous.
constexpr int TP_DMAX = 9;
template <typename T, int D3>
__device__ __forceinline__ void tp_tied_z(const T* a, const T* b, const T* __restrict__ C, int d1, int d2, T (&z)[TP_DMAX]) {
  for (int i = 0; i < d1; ++i) {
    const T ai = a[i];
    for (int j = 0; j < d2; ++j) {
      const T ab = ai * b[j];
      const T* c = C + (i * d2 + j) * D3;
#pragma unroll
      for (int k = 0; k < D3; ++k) z[k] += c[k] * ab;
    }
  }
}
template <typename T>
__device__ __forceinline__ T tp_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T>
__global__ void __launch_bounds__(256) k_tp_tied(TpArgsF a, const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ cg,
                                                 const T* __restrict__ weight, T* __restrict__ out) {
  __shared__ T stage[4][2][64 * TP_DMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= a.n) return;
  const T* A = x1 + n * a.dim1;
  const T* B = x2 + n * a.dim2;
  T* sa = stage[wave][0] + lane * TP_DMAX;
  T* sb = stage[wave][1] + lane * TP_DMAX;
  for (int si = 0; si < a.n_slots; ++si) {
    const TpSlot sl = a.slot[si];
    const bool reduce = a.path[sl.p0].mode == TP_UUW;
    const int mul_u = sl.mulo;     // (tied output) the index the lanes walk
    if (reduce) {   // out[w, k] = sum_paths coeff sum_u W[u, w] z_u[k]: per lane partial sums over its u, then the wave's sum
      for (int w0 = 0; w0 < sl.mulo; ++w0) {
        T part[TP_DMAX];
#pragma unroll
        for (int k = 0; k < TP_DMAX; ++k) part[k] = T(0);
        for (int pi = sl.p0; pi < sl.p1; ++pi) {
          const TpPathF p = a.path[pi];
          for (int u = lane; u < p.mul1; u += 64) {   // (the paths into one block may sum over different multiplicities)
            for (int i = 0; i < p.d1; ++i) sa[i] = A[p.off1 + u * p.d1 + i];
            for (int j = 0; j < p.d2; ++j) sb[j] = B[p.off2 + u * p.d2 + j];
            T z[TP_DMAX];
#pragma unroll
            for (int k = 0; k < TP_DMAX; ++k) z[k] = T(0);
            const T* C = cg + p.cg_off;
            switch (sl.d3) {
              case 1: tp_tied_z<T, 1>(sa, sb, C, p.d1, p.d2, z); break;
              case 3: tp_tied_z<T, 3>(sa, sb, C, p.d1, p.d2, z); break;
              case 5: tp_tied_z<T, 5>(sa, sb, C, p.d1, p.d2, z); break;
              case 7: tp_tied_z<T, 7>(sa, sb, C, p.d1, p.d2, z); break;
              default: tp_tied_z<T, 9>(sa, sb, C, p.d1, p.d2, z);
            }
            const T wv = (T)p.coeff * (p.w_off >= 0 ? weight[n * a.w_stride + p.w_off + u * sl.mulo + w0] : T(1));
#pragma unroll
            for (int k = 0; k < TP_DMAX; ++k) part[k] += wv * z[k];
          }
        }
#pragma unroll
        for (int k = 0; k < TP_DMAX; ++k)
          if (k < sl.d3) {
            const T tot = tp_wave_sum(part[k]);
            if (lane == 0) out[n * a.dim_out + sl.offo + w0 * sl.d3 + k] = tot;
          }
      }
      continue;
    }
    for (int u = lane; u < mul_u; u += 64) {   // output index tied to the lane: 'uuu', 'uvu', 'uvv'
      T acc[TP_DMAX];
#pragma unroll
      for (int k = 0; k < TP_DMAX; ++k) acc[k] = T(0);
      for (int pi = sl.p0; pi < sl.p1; ++pi) {
        const TpPathF p = a.path[pi];
        const int loops = p.mode == TP_UVU ? p.mul2 : (p.mode == TP_UVV ? p.mul1 : 1);
        const T* C = cg + p.cg_off;
        const T* W = p.w_off >= 0 ? weight + n * a.w_stride + p.w_off : nullptr;
        for (int t = 0; t < loops; ++t) {
          const int ua = p.mode == TP_UVV ? t : u, ub = p.mode == TP_UVU ? t : u;
          for (int i = 0; i < p.d1; ++i) sa[i] = A[p.off1 + ua * p.d1 + i];
          for (int j = 0; j < p.d2; ++j) sb[j] = B[p.off2 + ub * p.d2 + j];
          T z[TP_DMAX];
#pragma unroll
          for (int k = 0; k < TP_DMAX; ++k) z[k] = T(0);
          switch (sl.d3) {
            case 1: tp_tied_z<T, 1>(sa, sb, C, p.d1, p.d2, z); break;
            case 3: tp_tied_z<T, 3>(sa, sb, C, p.d1, p.d2, z); break;
            case 5: tp_tied_z<T, 5>(sa, sb, C, p.d1, p.d2, z); break;
            case 7: tp_tied_z<T, 7>(sa, sb, C, p.d1, p.d2, z); break;
            default: tp_tied_z<T, 9>(sa, sb, C, p.d1, p.d2, z);
          }
          T wv = (T)p.coeff;
          if (W) wv *= p.mode == TP_UUU ? W[u] : (p.mode == TP_UVU ? W[u * p.mul2 + t] : W[t * p.mul2 + u]);
#pragma unroll
          for (int k = 0; k < TP_DMAX; ++k) acc[k] += wv * z[k];
        }
      }
      T* o = out + n * a.dim_out + sl.offo + u * sl.d3;
#pragma unroll
      for (int k = 0; k < TP_DMAX; ++k)
        if (k < sl.d3) o[k] = acc[k];
    }
  }
}

// Weight gradients of every weighted path in one launch: dW[e] = coeff sum_n sum_k g[n, w, k] pair(u, v, k) for the weight element e =
// (path, u, v, w as the mode has them).  Shared weights: a workgroup sums a chunk of nodes for a run of 256 weight elements and writes
// parts[chunk][e] (the caller adds the chunks in order: deterministic); per-sample weights: one node per "chunk", written straight to
// dW[n][e].
struct TpWPath {
  int off1, off2, offo;
  int mul1, mul2, mulo;
  int d1, d2, d3;
  int mode, cg_off;
  int w_off, w_numel;
  double coeff;
};
struct TpWArgs {
  int64_t n;
  int dim1, dim2, dim_out;
  int n_paths, w_total, chunk;   // nodes per chunk
  TpWPath path[TP_MAXP];
};
template <typename T>
__global__ void __launch_bounds__(256) k_tp_wgrad(TpWArgs a, const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ g,
                                                  const T* __restrict__ cg, T* __restrict__ parts) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.w_total) return;
  int pi = 0;
  while (pi + 1 < a.n_paths && e >= a.path[pi + 1].w_off) ++pi;
  const TpWPath p = a.path[pi];
  const int le = e - p.w_off;
  int u, v, w;
  switch (p.mode) {
    case TP_UVW: u = le / (p.mul2 * p.mulo); v = (le / p.mulo) % p.mul2; w = le % p.mulo; break;
    case TP_UVU: u = le / p.mul2; v = le % p.mul2; w = u; break;
    case TP_UVV: u = le / p.mul2; v = le % p.mul2; w = v; break;
    case TP_UUW: u = le / p.mulo; v = u; w = le % p.mulo; break;
    case TP_UUU: u = v = w = le; break;
    default: u = le / p.mul2; v = le % p.mul2; w = le;   // TP_UVUV
  }
  const T* C = cg + p.cg_off;
  const int64_t c = blockIdx.y, n0 = c * a.chunk, n1 = min(a.n, n0 + a.chunk);
  T acc = T(0);
  for (int64_t n = n0; n < n1; ++n) {
    const T* A = x1 + n * a.dim1 + p.off1 + u * p.d1;
    const T* B = x2 + n * a.dim2 + p.off2 + v * p.d2;
    const T* G = g + n * a.dim_out + p.offo + w * p.d3;
    T s = T(0);
    for (int i = 0; i < p.d1; ++i) {
      const T ai = A[i];
      for (int j = 0; j < p.d2; ++j) {
        const T ab = ai * B[j];
        for (int k = 0; k < p.d3; ++k) s += C[(i * p.d2 + j) * p.d3 + k] * ab * G[k];
      }
    }
    acc += s;
  }
  parts[c * a.w_total + e] = (T)p.coeff * acc;
}

}  // namespace xeq

using namespace xeq;

extern "C" {

int xeq_tensor_product_path(int dtype, const void* x1, const void* x2, int64_t n, int dim1, int dim2, int dim_out, int off1,
                            int off2, int off_out, int mul1, int mul2, int mul_out, int l1, int l2, int l3, int mode,
                            const void* cg, const void* weight, int64_t weight_stride, double coeff, void* out, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && mul1 > 0 && mul2 > 0 && mul_out > 0 && l1 >= 0 && l2 >= 0 && l3 >= 0 && l1 <= 8 && l2 <= 8 && l3 <= 8,
                "xeq_tensor_product_path: bad sizes");
  XEQ_CHECK_ARG(mode >= TP_UVW && mode <= TP_UVUV, "xeq_tensor_product_path: unknown connection mode %d", mode);
  XEQ_CHECK_ARG(l3 >= (l1 > l2 ? l1 - l2 : l2 - l1) && l3 <= l1 + l2, "xeq_tensor_product_path: (%d, %d, %d) violates the triangle rule", l1, l2, l3);
  const bool tie_uv = mode == TP_UUW || mode == TP_UUU;
  XEQ_CHECK_ARG(!tie_uv || mul1 == mul2, "xeq_tensor_product_path: mode needs equal input multiplicities");
  XEQ_CHECK_ARG((mode != TP_UUU && mode != TP_UVU) || mul_out == mul1, "xeq_tensor_product_path: output multiplicity must equal mul1");
  XEQ_CHECK_ARG(mode != TP_UVV || mul_out == mul2, "xeq_tensor_product_path: output multiplicity must equal mul2");
  XEQ_CHECK_ARG(mode != TP_UVUV || mul_out == mul1 * mul2, "xeq_tensor_product_path: output multiplicity must equal mul1 * mul2");
  XEQ_CHECK_ARG(off1 + mul1 * (2 * l1 + 1) <= dim1 && off2 + mul2 * (2 * l2 + 1) <= dim2 && off_out + mul_out * (2 * l3 + 1) <= dim_out,
                "xeq_tensor_product_path: irrep block outside its row");
  if (n == 0) return XEQ_OK;
  TpPath p{n, dim1, dim2, dim_out, off1, off2, off_out, mul1, mul2, mul_out, 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1, mode,
           weight != nullptr ? 1 : 0, weight_stride};
  const int64_t total = n * (int64_t)mul_out * p.d3;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_tp_path<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, (const T*)x1,
                       (const T*)x2, (const T*)cg, (const T*)weight, (T)coeff, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_tensor_product_path");
  return XEQ_OK;
}


/* All instructions of a product in one launch (see include/xeq.h).  paths: n_paths x 10 ints (off1, off2, off_out, mul1, mul2, mul_out,
 * l1, l2, l3, mode), sorted by off_out (paths of one output block adjacent, in instruction order); cg: the paths' 3j tables one after the
 * other (device memory, dtype of x; cg_off[p] = first element); w_off[p] = first weight of path p or -1; coeff[p].  Overwrites the
 * output blocks the paths end in (the caller zero-fills `out` once if some block has no path). */
int xeq_tensor_product(int dtype, const void* x1, const void* x2, int64_t n, int dim1, int dim2, int dim_out, int n_paths,
                       const int32_t* paths, const void* cg, const int32_t* cg_off, int cg_floats, const void* weight, int64_t weight_stride,
                       const int32_t* w_off, const double* coeff, void* out, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n_paths > 0 && n_paths <= TP_MAXP && paths && cg && cg_off && w_off && coeff && dim1 > 0 && dim2 > 0 && dim_out > 0,
                "xeq_tensor_product: bad arguments (at most %d paths per launch)", TP_MAXP);
  if (n == 0) return XEQ_OK;
  XEQ_CHECK_ARG(x1 && x2 && out, "xeq_tensor_product: null buffer");
  TpArgsF a;
  a.n = n; a.dim1 = dim1; a.dim2 = dim2; a.dim_out = dim_out; a.w_stride = weight_stride; a.cg_floats = cg_floats; a.n_slots = 0;
  bool any_w = false;
  for (int p = 0; p < n_paths; ++p) {
    const int32_t* q = paths + 10 * p;
    const int off1 = q[0], off2 = q[1], offo = q[2], m1 = q[3], m2 = q[4], mo = q[5], l1 = q[6], l2 = q[7], l3 = q[8], mode = q[9];
    XEQ_CHECK_ARG(m1 > 0 && m2 > 0 && mo > 0 && l1 >= 0 && l2 >= 0 && l3 >= 0 && l1 <= 8 && l2 <= 8 && l3 <= 8, "xeq_tensor_product: bad sizes (path %d)", p);
    XEQ_CHECK_ARG(mode >= TP_UVW && mode <= TP_UVUV, "xeq_tensor_product: unknown connection mode %d", mode);
    XEQ_CHECK_ARG(l3 >= (l1 > l2 ? l1 - l2 : l2 - l1) && l3 <= l1 + l2, "xeq_tensor_product: (%d, %d, %d) violates the triangle rule", l1, l2, l3);
    XEQ_CHECK_ARG(!(mode == TP_UUW || mode == TP_UUU) || m1 == m2, "xeq_tensor_product: mode needs equal input multiplicities");
    XEQ_CHECK_ARG((mode != TP_UUU && mode != TP_UVU) || mo == m1, "xeq_tensor_product: output multiplicity must equal mul1");
    XEQ_CHECK_ARG(mode != TP_UVV || mo == m2, "xeq_tensor_product: output multiplicity must equal mul2");
    XEQ_CHECK_ARG(mode != TP_UVUV || mo == m1 * m2, "xeq_tensor_product: output multiplicity must equal mul1 * mul2");
    XEQ_CHECK_ARG(off1 >= 0 && off2 >= 0 && offo >= 0 && off1 + m1 * (2 * l1 + 1) <= dim1 && off2 + m2 * (2 * l2 + 1) <= dim2 && offo + mo * (2 * l3 + 1) <= dim_out,
                  "xeq_tensor_product: irrep block outside its row");
    XEQ_CHECK_ARG(cg_off[p] >= 0 && cg_off[p] + (2 * l1 + 1) * (2 * l2 + 1) * (2 * l3 + 1) <= cg_floats, "xeq_tensor_product: 3j table outside its buffer");
    if (a.n_slots == 0 || a.slot[a.n_slots - 1].offo != offo) {
      XEQ_CHECK_ARG(a.n_slots < TP_MAXS && (a.n_slots == 0 || a.slot[a.n_slots - 1].offo < offo), "xeq_tensor_product: paths must be sorted by output block (at most %d blocks)", TP_MAXS);
      a.slot[a.n_slots] = TpSlot{offo, mo, 2 * l3 + 1, 0, p, p};
      ++a.n_slots;
    }
    TpSlot& sl = a.slot[a.n_slots - 1];
    XEQ_CHECK_ARG(sl.mulo == mo && sl.d3 == 2 * l3 + 1, "xeq_tensor_product: paths into one output block disagree on its shape");
    sl.p1 = p + 1;
    if (mode == TP_UVW || mode == TP_UUW) sl.blocked = 1;
    a.path[p] = TpPathF{off1, off2, m1, m2, 2 * l1 + 1, 2 * l2 + 1, mode, cg_off[p], w_off[p], coeff[p]};
    any_w = any_w || w_off[p] >= 0;
  }
  // a slot that mixes blocked and element-wise paths runs blocked: the element-wise modes then only fill acc[0] of w0 = wb * TP_WB -- not
  // what they need, so such a slot is refused (the instruction builders never produce one: a product has one connection mode)
  for (int s = 0; s < a.n_slots; ++s)
    for (int p = a.slot[s].p0; p < a.slot[s].p1; ++p)
      XEQ_CHECK_ARG(!a.slot[s].blocked || a.path[p].mode == TP_UVW || a.path[p].mode == TP_UUW, "xeq_tensor_product: an output block mixes 'uvw' / 'uuw' with other modes");
  XEQ_CHECK_ARG(!any_w || weight, "xeq_tensor_product: weighted paths need the weight buffer");
  bool tied = getenv("XEQ_TP_GENERIC") == nullptr;   // (development: XEQ_TP_GENERIC forces the element-per-thread form)
  for (int s = 0; s < a.n_slots && tied; ++s)
    for (int p = a.slot[s].p0; p < a.slot[s].p1; ++p) {
      const int m = a.path[p].mode;
      tied = tied && (m == TP_UUU || m == TP_UUW || m == TP_UVU || m == TP_UVV) && ((m == TP_UUW) == (a.path[a.slot[s].p0].mode == TP_UUW)) &&
             a.path[p].d1 <= TP_DMAX && a.path[p].d2 <= TP_DMAX && a.slot[s].d3 <= TP_DMAX;
    }
  if (tied) {
    XEQ_DISPATCH_FLOAT(dtype, {
      hipLaunchKernelGGL((k_tp_tied<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, (const T*)x1, (const T*)x2, (const T*)cg,
                         (const T*)weight, (T*)out);
    });
    XEQ_CHECK_LAUNCH("xeq_tensor_product");
    return XEQ_OK;
  }
  const size_t esz = dtype == XEQ_F64 ? 8 : 4;
  const size_t row = (size_t)(dim1 + dim2) * esz, fixed = (size_t)cg_floats * esz;
  XEQ_CHECK_ARG(fixed + row <= 60000, "xeq_tensor_product: rows of %d + %d elements and %d table entries do not fit the LDS tile", dim1, dim2, cg_floats);
  int tn = (int)((60000 - fixed) / row);
  tn = tn > 16 ? 16 : tn;
  // enough workgroups to fill the chip before tiles grow
  while (tn > 1 && (n + tn - 1) / tn < 1024) tn = (tn + 1) / 2;
  a.tn = tn;
  const size_t lds = fixed + (size_t)tn * row;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_tp_fused<T>), dim3((unsigned)((n + tn - 1) / tn)), dim3(256), lds, (hipStream_t)stream, a, (const T*)x1, (const T*)x2,
                       (const T*)cg, (const T*)weight, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_tensor_product");
  return XEQ_OK;
}

/* Weight gradients of the weighted paths of a product (same path table; w_off[p] >= 0 and w_numel[p] for weighted paths, in increasing
 * w_off order; w_total = all weights).  shared = 1: parts [n_chunks, w_total] with n_chunks = xeq_tensor_product_wgrad_chunks(n) (the caller
 * sums them in order); shared = 0: parts = dW [n, w_total]. */
int64_t xeq_tensor_product_wgrad_chunks(int64_t n) { return n <= 0 ? 0 : (n + 63) / 64 < 512 ? (n + 63) / 64 : 512; }
int xeq_tensor_product_wgrad(int dtype, const void* x1, const void* x2, const void* g, int64_t n, int dim1, int dim2, int dim_out, int n_paths,
                             const int32_t* paths, const void* cg, const int32_t* cg_off, const int32_t* w_off, const int32_t* w_numel, int w_total,
                             const double* coeff, int shared, void* parts, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n_paths > 0 && n_paths <= TP_MAXP && paths && cg && cg_off && w_off && w_numel && coeff && w_total > 0,
                "xeq_tensor_product_wgrad: bad arguments");
  if (n == 0) return XEQ_OK;
  XEQ_CHECK_ARG(x1 && x2 && g && parts, "xeq_tensor_product_wgrad: null buffer");
  TpWArgs a;
  a.n = n; a.dim1 = dim1; a.dim2 = dim2; a.dim_out = dim_out; a.w_total = w_total; a.n_paths = 0;
  int expect = 0;
  for (int p = 0; p < n_paths; ++p) {
    if (w_off[p] < 0) continue;
    const int32_t* q = paths + 10 * p;
    XEQ_CHECK_ARG(w_off[p] == expect, "xeq_tensor_product_wgrad: weighted paths must tile the weight vector in order");
    expect += w_numel[p];
    a.path[a.n_paths++] = TpWPath{q[0], q[1], q[2], q[3], q[4], q[5], 2 * q[6] + 1, 2 * q[7] + 1, 2 * q[8] + 1, q[9], cg_off[p], w_off[p], w_numel[p], coeff[p]};
  }
  XEQ_CHECK_ARG(expect == w_total && a.n_paths > 0, "xeq_tensor_product_wgrad: weights of the paths do not add up to w_total");
  const int64_t chunks = shared ? xeq_tensor_product_wgrad_chunks(n) : n;
  a.chunk = shared ? (int)((n + chunks - 1) / chunks) : 1;
  XEQ_CHECK_ARG(chunks <= 65535 * 1 || !shared, "xeq_tensor_product_wgrad: too many chunks");
  for (int64_t c0 = 0; c0 < chunks; c0 += 65535) {   // gridDim.y limit
    const int64_t cn = chunks - c0 < 65535 ? chunks - c0 : 65535;
    TpWArgs b = a;
    b.n = n;
    XEQ_DISPATCH_FLOAT(dtype, {
      const T* X1 = (const T*)x1 + c0 * a.chunk * dim1;
      const T* X2 = (const T*)x2 + c0 * a.chunk * dim2;
      const T* G = (const T*)g + c0 * a.chunk * dim_out;
      b.n = n - c0 * a.chunk;
      hipLaunchKernelGGL((k_tp_wgrad<T>), dim3((unsigned)((w_total + 255) / 256), (unsigned)cn), dim3(256), 0, (hipStream_t)stream, b, X1, X2, G,
                         (const T*)cg, (T*)parts + c0 * w_total);
    });
  }
  XEQ_CHECK_LAUNCH("xeq_tensor_product_wgrad");
  return XEQ_OK;
}

}  // extern "C"
