// Node-side kernels of a TRAINING pass that is differentiated twice (a loss on forces: nn/basic.py:143-159 with create_graph=training,
// utils/trainer.py:295-308).  Three per-node functions of the XPaiNN blocks, each as a forward kernel and a reverse kernel:
//   norm        (s, x) -> (LayerNorm(s), EquivariantLayerNorm(x))                      nn/xpainn.py:130-131, :213-214; nn/o3layer.py:145-171
//   uv          (U, V) -> (Invariant(V) = sqrt(sum_m V^2 + eps^2) - eps, sum_m U V)    nn/xpainn.py:218-224; nn/o3layer.py:40-44, :61-68
//   out         (U, a, inner) -> (a_sv inner + a_ss, U a_vv)                          nn/xpainn.py:226-230
// The SECOND order comes from the same two kernel bodies evaluated on dual numbers (value + tangent): with f the function, g the
// cotangent of its output and u the cotangent that arrives for the first-order gradient xbar = J(x)^T g,
//   d<u, xbar>/dg     = J(x) u                 = tangent of the FORWARD body at x + eps u,
//   d<u, xbar>/dx     = sum_k g_k Hess f_k u   = tangent of the REVERSE body's input gradient at x + eps u   (Hessians are symmetric),
//   d<u, xbar>/dtheta                          = tangent of the REVERSE body's parameter gradient at x + eps u,
// so no second-order formula is written by hand.  Host side: ops.NormFn / UvFn / UpdateOutFn and their *Grad nodes.
// f32 and f64; these kernels are plain (a wave per node for the norms, a thread per (node, channel) for the other two): the pass they
// serve is bound by the edge kernels and the library GEMMs around them.
#include "xeq_common.h"

namespace xeq {
namespace tn {

template <typename T>
struct Dual {
  T v, d;
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ Dual(T v_) : v(v_), d(T(0)) {}
  __device__ __forceinline__ Dual(T v_, T d_) : v(v_), d(d_) {}
};
#define XEQ_TN_OP __device__ __forceinline__
template <typename T> XEQ_TN_OP Dual<T> operator+(Dual<T> a, Dual<T> b) { return {a.v + b.v, a.d + b.d}; }
template <typename T> XEQ_TN_OP Dual<T> operator-(Dual<T> a, Dual<T> b) { return {a.v - b.v, a.d - b.d}; }
template <typename T> XEQ_TN_OP Dual<T> operator-(Dual<T> a) { return {-a.v, -a.d}; }
template <typename T> XEQ_TN_OP Dual<T> operator*(Dual<T> a, Dual<T> b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
template <typename T> XEQ_TN_OP Dual<T> operator/(Dual<T> a, Dual<T> b) {
  const T q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
template <typename T> XEQ_TN_OP Dual<T> operator+(Dual<T> a, T b) { return {a.v + b, a.d}; }
template <typename T> XEQ_TN_OP Dual<T> operator-(Dual<T> a, T b) { return {a.v - b, a.d}; }
template <typename T> XEQ_TN_OP Dual<T> operator*(Dual<T> a, T b) { return {a.v * b, a.d * b}; }
template <typename T> XEQ_TN_OP Dual<T> operator*(T b, Dual<T> a) { return {a.v * b, a.d * b}; }
template <typename T> XEQ_TN_OP Dual<T> operator/(Dual<T> a, T b) { return {a.v / b, a.d / b}; }
template <typename T> XEQ_TN_OP Dual<T>& operator+=(Dual<T>& a, Dual<T> b) {
  a.v += b.v;
  a.d += b.d;
  return a;
}

XEQ_TN_OP float rsqrt_s(float x) { return 1.0f / sqrtf(x); }
XEQ_TN_OP double rsqrt_s(double x) { return 1.0 / sqrt(x); }
template <typename T> XEQ_TN_OP Dual<T> rsqrt_s(Dual<T> x) {
  const T r = rsqrt_s(x.v);
  return {r, T(-0.5) * r / x.v * x.d};
}
XEQ_TN_OP float sqrt_s(float x) { return sqrtf(x); }
XEQ_TN_OP double sqrt_s(double x) { return sqrt(x); }
template <typename T> XEQ_TN_OP Dual<T> sqrt_s(Dual<T> x) {
  const T r = sqrt_s(x.v);
  return {r, x.d / (T(2) * r)};
}
XEQ_TN_OP float wsum(float v) { return wave_sum<float>(v); }
XEQ_TN_OP double wsum(double v) { return wave_sum<double>(v); }
template <typename T> XEQ_TN_OP Dual<T> wsum(Dual<T> a) { return {wave_sum<T>(a.v), wave_sum<T>(a.d)}; }

// value (+ tangent) in, value or tangent out
template <typename T, bool DUAL>
struct Acc;
template <typename T>
struct Acc<T, false> {
  using S = T;
  static XEQ_TN_OP S ld(const T* __restrict__ p, const T* __restrict__, int64_t i) { return p[i]; }
  static XEQ_TN_OP void st(T* __restrict__ o, int64_t i, S v) { o[i] = v; }
};
template <typename T>
struct Acc<T, true> {
  using S = Dual<T>;
  static XEQ_TN_OP S ld(const T* __restrict__ p, const T* __restrict__ t, int64_t i) { return S(p[i], t[i]); }
  static XEQ_TN_OP void st(T* __restrict__ o, int64_t i, S v) { o[i] = v.d; }
};

// element i of an e3nn row -> channel, component, l, and its place in the layout (0: the e3nn row, 1: BT = per l a row-major
// [N (2l+1), mul_l] matrix, the matrices one after the other)
struct Elem {
  int ch, m, l;
};
__device__ __forceinline__ Elem elem_of(const Irreps& ir, int i) {
  Elem e;
  const int m0 = ir.mul[0], m1 = ir.mul[1];
  if (i < m0) {
    e.ch = i;
    e.m = 0;
    e.l = 0;
  } else if (i < m0 + 3 * m1) {
    const int r = i - m0;
    e.ch = m0 + r / 3;
    e.m = r % 3;
    e.l = 1;
  } else {
    const int r = i - m0 - 3 * m1;
    e.ch = m0 + m1 + r / 5;
    e.m = r % 5;
    e.l = 2;
  }
  return e;
}
__device__ __forceinline__ int64_t place(const Irreps& ir, int64_t N, int64_t n, int i, const Elem& e, int layout) {
  if (layout == 0) return n * ir.D() + i;
  const int64_t base = e.l == 0 ? 0 : (e.l == 1 ? (int64_t)ir.mul[0] : (int64_t)ir.mul[0] + 3 * ir.mul[1]);
  const int up = e.ch - (e.l == 0 ? 0 : (e.l == 1 ? ir.mul[0] : ir.mul[0] + ir.mul[1]));
  return N * base + (n * (2 * e.l + 1) + e.m) * ir.mul[e.l] + up;
}

template <typename T>
struct NormArgs {
  int64_t N;
  const T *s, *s_t, *x, *x_t, *ln_w, *ln_b, *eq_w, *eq_b, *g_s, *g_x;
  T *o_s, *o_x, *rows;
  int F, layout;
  Irreps ir;
  T eps_ln, eps_eq;
};

template <typename T, bool DUAL>
__global__ void __launch_bounds__(256) k_norm_fwd(NormArgs<T> a) {
  using A = Acc<T, DUAL>;
  using S = typename A::S;
  const int lane = threadIdx.x & 63;
  const int F = a.F, D = a.ir.D(), C = a.ir.C(), M0 = a.ir.mul[0];
  for (int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); n < a.N; n += (int64_t)gridDim.x * 4) {
    S acc = S(T(0));
    for (int i = lane; i < F; i += 64) acc += A::ld(a.s, a.s_t, n * F + i);
    const S mu = wsum(acc) * (T(1) / T(F));
    acc = S(T(0));
    for (int i = lane; i < F; i += 64) {
      const S c = A::ld(a.s, a.s_t, n * F + i) - mu;
      acc += c * c;
    }
    const S r = rsqrt_s(wsum(acc) * (T(1) / T(F)) + a.eps_ln);
    for (int i = lane; i < F; i += 64) {
      const S z = (A::ld(a.s, a.s_t, n * F + i) - mu) * r;
      A::st(a.o_s, n * F + i, z * a.ln_w[i] + a.ln_b[i]);
    }
    acc = S(T(0));
    for (int i = lane; i < M0; i += 64) acc += A::ld(a.x, a.x_t, n * D + i);
    const S m0 = M0 > 0 ? wsum(acc) * (T(1) / T(M0 > 0 ? M0 : 1)) : S(T(0));
    acc = S(T(0));
    for (int i = lane; i < D; i += 64) {
      S c = A::ld(a.x, a.x_t, n * D + i);
      if (i < M0) c = c - m0;
      acc += c * c;
    }
    const S inv = rsqrt_s(wsum(acc) * (T(1) / T(C)) + a.eps_eq);
    for (int i = lane; i < D; i += 64) {
      S c = A::ld(a.x, a.x_t, n * D + i);
      if (i < M0) c = c - m0;
      const Elem e = elem_of(a.ir, i);
      S y = c * inv * a.eq_w[e.ch];
      if (i < M0) y = y + a.eq_b[i];
      A::st(a.o_x, place(a.ir, a.N, n, i, e, a.layout), y);
    }
  }
}

// o_s / o_x: dL/ds, dL/dx (e3nn rows); rows[n] = [d ln_w (F) | d ln_b (F) | d eq_w (C) | d eq_b (mul0)] of the node
template <typename T, bool DUAL>
__global__ void __launch_bounds__(256) k_norm_bwd(NormArgs<T> a) {
  using A = Acc<T, DUAL>;
  using S = typename A::S;
  const int lane = threadIdx.x & 63;
  const int F = a.F, D = a.ir.D(), C = a.ir.C(), M0 = a.ir.mul[0];
  const int RW = 2 * F + C + M0;
  for (int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); n < a.N; n += (int64_t)gridDim.x * 4) {
    T* row = a.rows + n * RW;
    S acc = S(T(0));
    for (int i = lane; i < F; i += 64) acc += A::ld(a.s, a.s_t, n * F + i);
    const S mu = wsum(acc) * (T(1) / T(F));
    acc = S(T(0));
    for (int i = lane; i < F; i += 64) {
      const S c = A::ld(a.s, a.s_t, n * F + i) - mu;
      acc += c * c;
    }
    const S r = rsqrt_s(wsum(acc) * (T(1) / T(F)) + a.eps_ln);
    S sa = S(T(0)), saz = S(T(0));
    for (int i = lane; i < F; i += 64) {
      const S z = (A::ld(a.s, a.s_t, n * F + i) - mu) * r;
      const T ai = a.g_s[n * F + i] * a.ln_w[i];
      sa += S(ai);
      saz += z * ai;
    }
    sa = wsum(sa) * (T(1) / T(F));
    saz = wsum(saz) * (T(1) / T(F));
    for (int i = lane; i < F; i += 64) {
      const S z = (A::ld(a.s, a.s_t, n * F + i) - mu) * r;
      const T g = a.g_s[n * F + i];
      A::st(a.o_s, n * F + i, r * (S(g * a.ln_w[i]) - sa - z * saz));
      A::st(row, i, z * g);
      A::st(row, F + i, S(g));
    }
    acc = S(T(0));
    for (int i = lane; i < M0; i += 64) acc += A::ld(a.x, a.x_t, n * D + i);
    const S m0 = M0 > 0 ? wsum(acc) * (T(1) / T(M0 > 0 ? M0 : 1)) : S(T(0));
    acc = S(T(0));
    S At = S(T(0));
    for (int i = lane; i < D; i += 64) {
      S c = A::ld(a.x, a.x_t, n * D + i);
      if (i < M0) c = c - m0;
      acc += c * c;
      const Elem e = elem_of(a.ir, i);
      At += c * (a.g_x[place(a.ir, a.N, n, i, e, a.layout)] * a.eq_w[e.ch]);
    }
    const S inv = rsqrt_s(wsum(acc) * (T(1) / T(C)) + a.eps_eq);
    At = wsum(At);
    const S k3 = inv * inv * inv * At * (T(1) / T(C));
    S c0 = S(T(0));   // sum of dL/dc over the scalars
    for (int i = lane; i < M0; i += 64) {
      const S c = A::ld(a.x, a.x_t, n * D + i) - m0;
      const T t = a.g_x[place(a.ir, a.N, n, i, elem_of(a.ir, i), a.layout)] * a.eq_w[i];
      c0 += inv * t - k3 * c;
    }
    c0 = M0 > 0 ? wsum(c0) * (T(1) / T(M0 > 0 ? M0 : 1)) : S(T(0));
    for (int i = lane; i < D; i += 64) {
      S c = A::ld(a.x, a.x_t, n * D + i);
      if (i < M0) c = c - m0;
      const Elem e = elem_of(a.ir, i);
      const T g = a.g_x[place(a.ir, a.N, n, i, e, a.layout)];
      S cb = inv * (g * a.eq_w[e.ch]) - k3 * c;
      if (i < M0) {
        cb = cb - c0;
        A::st(row, 2 * F + C + i, S(g));
      }
      A::st(a.o_x, n * D + i, cb);
    }
    for (int ch = lane; ch < C; ch += 64) {   // d eq_w: the channel's components
      int l, off;
      a.ir.locate(ch, l, off);
      S w = S(T(0));
      for (int m = 0; m < 2 * l + 1; ++m) {
        const int i = off + m;
        S c = A::ld(a.x, a.x_t, n * D + i);
        if (i < M0) c = c - m0;
        Elem e;
        e.ch = ch;
        e.m = m;
        e.l = l;
        w += c * a.g_x[place(a.ir, a.N, n, i, e, a.layout)];
      }
      A::st(row, 2 * F + ch, w * inv);
    }
  }
}

// uv_l: [N (2l+1), 2 mul_l] row-major, U in the first mul_l columns, V behind them
template <typename T>
struct UvArgs {
  int64_t N;
  const T* uv[3];
  const T* uv_t[3];
  const T *a, *a_t, *inner, *inner_t;   // out: a [N, C + 2F] = [a_vv | a_sv | a_ss], inner [N, F]
  const T *g0, *g1;                     // uv: g0 = dL/d[vn | dot] [N, 2C];  out: g0 = dL/d(delta s) [N, F], g1 = dL/d(delta x) [N, D]
  T* o_uv[3];
  T *o0, *o1;                           // uv fwd: o0 = [vn | dot];  out fwd: o0 = delta s, o1 = delta x;  out bwd: o0 = dL/da, o1 = dL/dinner
  int F;
  Irreps ir;
  T eps;
};

template <typename T>
__device__ __forceinline__ int64_t uv_at(const Irreps& ir, int64_t n, int l, int up, int m) {
  return (n * (2 * l + 1) + m) * (2 * (int64_t)ir.mul[l]) + up;
}

template <typename T, bool DUAL, bool BWD>
__global__ void __launch_bounds__(256) k_uv(UvArgs<T> a) {
  using A = Acc<T, DUAL>;
  using S = typename A::S;
  const int C = a.ir.C();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / C;
  if (n >= a.N) return;
  const int ch = (int)(idx - n * C);
  int l, off;
  a.ir.locate(ch, l, off);
  const int up = ch - (l == 0 ? 0 : (l == 1 ? a.ir.mul[0] : a.ir.mul[0] + a.ir.mul[1]));
  const int mul = a.ir.mul[l];
  S U[5], V[5];
  S vv = S(a.eps * a.eps), dot = S(T(0));
  for (int m = 0; m < 2 * l + 1; ++m) {
    const int64_t p = uv_at<T>(a.ir, n, l, up, m);
    U[m] = A::ld(a.uv[l], a.uv_t[l], p);
    V[m] = A::ld(a.uv[l], a.uv_t[l], p + mul);
    vv += V[m] * V[m];
    dot += U[m] * V[m];
  }
  const S nrm = sqrt_s(vv);
  if (!BWD) {
    A::st(a.o0, n * 2 * C + ch, nrm - a.eps);
    A::st(a.o0, n * 2 * C + C + ch, dot);
  } else {
    const T gv = a.g0[n * 2 * C + ch], gd = a.g0[n * 2 * C + C + ch];
    const S k = S(gv) / nrm;
    for (int m = 0; m < 2 * l + 1; ++m) {
      const int64_t p = uv_at<T>(a.ir, n, l, up, m);
      A::st(a.o_uv[l], p, V[m] * gd);
      A::st(a.o_uv[l], p + mul, k * V[m] + U[m] * gd);
    }
  }
}

template <typename T, bool DUAL, bool BWD>
__global__ void __launch_bounds__(256) k_out(UvArgs<T> a) {
  using A = Acc<T, DUAL>;
  using S = typename A::S;
  const int C = a.ir.C(), F = a.F, D = a.ir.D(), AW = C + 2 * F;
  const int TP = C > F ? C : F;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / TP;
  if (n >= a.N) return;
  const int t = (int)(idx - n * TP);
  if (t < C) {
    int l, off;
    a.ir.locate(t, l, off);
    const int up = t - (l == 0 ? 0 : (l == 1 ? a.ir.mul[0] : a.ir.mul[0] + a.ir.mul[1]));
    const int mul = a.ir.mul[l];
    const S avv = A::ld(a.a, a.a_t, n * AW + t);
    S acc = S(T(0));
    for (int m = 0; m < 2 * l + 1; ++m) {
      const int64_t p = uv_at<T>(a.ir, n, l, up, m);
      const S U = A::ld(a.uv[l], a.uv_t[l], p);
      if (!BWD) {
        A::st(a.o1, n * D + off + m, U * avv);
      } else {
        const T g = a.g1[n * D + off + m];
        A::st(a.o_uv[l], p, avv * g);
        A::st(a.o_uv[l], p + mul, S(T(0)));
        acc += U * g;
      }
    }
    if (BWD) A::st(a.o0, n * AW + t, acc);
  }
  if (t < F) {
    const S asv = A::ld(a.a, a.a_t, n * AW + C + t), ass = A::ld(a.a, a.a_t, n * AW + C + F + t);
    const S in = A::ld(a.inner, a.inner_t, n * F + t);
    if (!BWD) {
      A::st(a.o0, n * F + t, asv * in + ass);
    } else {
      const T g = a.g0[n * F + t];
      A::st(a.o0, n * AW + C + t, in * g);
      A::st(a.o0, n * AW + C + F + t, S(g));
      A::st(a.o1, n * F + t, asv * g);
    }
  }
}

}  // namespace tn
}  // namespace xeq

using namespace xeq;
using namespace xeq::tn;

static int tn_sizes(const char* who, int64_t n, int node_dim, const int32_t mul[3], Irreps& ir) {
  XEQ_CHECK_ARG(n >= 0 && n < (1ll << 31) && node_dim >= 1 && node_dim <= 4096, "%s: bad sizes", who);
  for (int l = 0; l < 3; ++l) {
    XEQ_CHECK_ARG(mul[l] >= 0 && mul[l] <= 4096, "%s: bad multiplicity", who);
    ir.mul[l] = mul[l];
  }
  XEQ_CHECK_ARG(ir.C() >= 1, "%s: no channels", who);
  return XEQ_OK;
}

extern "C" {

int xeq_train_norm(int dtype, int reverse, int64_t n, const void* s, const void* s_tan, const void* x, const void* x_tan, const void* ln_w,
                   const void* ln_b, const void* eq_w, const void* eq_b, const void* g_s, const void* g_x, int node_dim,
                   const int32_t mul[3], double eps_ln, double eps_eq, int xhat_layout, void* out_s, void* out_x, void* rows, void* stream) {
  Irreps ir{};
  int rc = tn_sizes("xeq_train_norm", n, node_dim, mul, ir);
  if (rc != XEQ_OK) return rc;
  XEQ_CHECK_ARG((s_tan == nullptr) == (x_tan == nullptr), "xeq_train_norm: tangents of s and x come together");
  XEQ_CHECK_ARG(xhat_layout == 0 || xhat_layout == 1, "xeq_train_norm: layout %d", xhat_layout);
  XEQ_CHECK_ARG(!reverse || (g_s && g_x && rows), "xeq_train_norm: the reverse form needs g_s, g_x and rows");
  if (n == 0) return XEQ_OK;
  const bool dual = s_tan != nullptr;
  dim3 grid((unsigned)((n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096));
  XEQ_DISPATCH_FLOAT(dtype, {
    NormArgs<T> a{n, (const T*)s, (const T*)s_tan, (const T*)x, (const T*)x_tan, (const T*)ln_w, (const T*)ln_b, (const T*)eq_w,
                  (const T*)eq_b, (const T*)g_s, (const T*)g_x, (T*)out_s, (T*)out_x, (T*)rows, node_dim, xhat_layout, ir, (T)eps_ln, (T)eps_eq};
    if (!reverse) {
      if (dual) hipLaunchKernelGGL((k_norm_fwd<T, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((k_norm_fwd<T, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
      if (dual) hipLaunchKernelGGL((k_norm_bwd<T, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((k_norm_bwd<T, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
  });
  XEQ_CHECK_LAUNCH("xeq_train_norm");
  return XEQ_OK;
}

#define XEQ_TN_LAUNCH(KERNEL, total)                                                                                         \
  do {                                                                                                                       \
    dim3 grid((unsigned)(((total) + 255) / 256));                                                                            \
    if (!reverse) {                                                                                                          \
      if (dual) hipLaunchKernelGGL((KERNEL<T, true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);                    \
      else hipLaunchKernelGGL((KERNEL<T, false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);                        \
    } else {                                                                                                                 \
      if (dual) hipLaunchKernelGGL((KERNEL<T, true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);                     \
      else hipLaunchKernelGGL((KERNEL<T, false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);                         \
    }                                                                                                                        \
  } while (0)

int xeq_train_uv(int dtype, int reverse, int64_t n, const void* const uv[3], const void* const uv_tan[3], const void* g, const int32_t mul[3],
                 double eps, void* out, void* const d_uv[3], void* stream) {
  Irreps ir{};
  int rc = tn_sizes("xeq_train_uv", n, 1, mul, ir);
  if (rc != XEQ_OK) return rc;
  XEQ_CHECK_ARG(!reverse || g, "xeq_train_uv: the reverse form needs g");
  if (n == 0) return XEQ_OK;
  const bool dual = uv_tan != nullptr;
  XEQ_DISPATCH_FLOAT(dtype, {
    UvArgs<T> a{};
    a.N = n;
    for (int l = 0; l < 3; ++l) {
      a.uv[l] = (const T*)uv[l];
      a.uv_t[l] = dual ? (const T*)uv_tan[l] : nullptr;
      a.o_uv[l] = d_uv ? (T*)d_uv[l] : nullptr;
    }
    a.g0 = (const T*)g;
    a.o0 = (T*)out;
    a.ir = ir;
    a.eps = (T)eps;
    XEQ_TN_LAUNCH(k_uv, n * ir.C());
  });
  XEQ_CHECK_LAUNCH("xeq_train_uv");
  return XEQ_OK;
}

int xeq_train_out(int dtype, int reverse, int64_t n, const void* const uv[3], const void* const uv_tan[3], const void* a_, const void* a_tan,
                  const void* inner, const void* inner_tan, const void* g_s, const void* g_x, int node_dim, const int32_t mul[3],
                  void* out0, void* out1, void* const d_uv[3], void* stream) {
  Irreps ir{};
  int rc = tn_sizes("xeq_train_out", n, node_dim, mul, ir);
  if (rc != XEQ_OK) return rc;
  XEQ_CHECK_ARG(!reverse || (g_s && g_x && d_uv), "xeq_train_out: the reverse form needs g_s, g_x and d_uv");
  XEQ_CHECK_ARG((uv_tan == nullptr) == (a_tan == nullptr) && (a_tan == nullptr) == (inner_tan == nullptr),
                "xeq_train_out: the tangents come together");
  if (n == 0) return XEQ_OK;
  const bool dual = uv_tan != nullptr;
  XEQ_DISPATCH_FLOAT(dtype, {
    UvArgs<T> a{};
    a.N = n;
    for (int l = 0; l < 3; ++l) {
      a.uv[l] = (const T*)uv[l];
      a.uv_t[l] = dual ? (const T*)uv_tan[l] : nullptr;
      a.o_uv[l] = d_uv ? (T*)d_uv[l] : nullptr;
    }
    a.a = (const T*)a_;
    a.a_t = (const T*)a_tan;
    a.inner = (const T*)inner;
    a.inner_t = (const T*)inner_tan;
    a.g0 = (const T*)g_s;
    a.g1 = (const T*)g_x;
    a.o0 = (T*)out0;
    a.o1 = (T*)out1;
    a.F = node_dim;
    a.ir = ir;
    const int TP = ir.C() > node_dim ? ir.C() : node_dim;
    XEQ_TN_LAUNCH(k_out, n * TP);
  });
  XEQ_CHECK_LAUNCH("xeq_train_out");
  return XEQ_OK;
}

}  // extern "C"
