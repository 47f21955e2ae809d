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

}  // extern "C"
