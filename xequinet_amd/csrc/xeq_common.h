// Shared device/host helpers for libxeq_hip.so (gfx950 only).
//
// Math restated from the reference (citations relative to /root/reference/xequinet/):
//   radial basis   nn/rbf.py:134-152 (SphericalBesselj0), :114-131 (GaussianSmearing)
//   envelopes      nn/rbf.py:43-57 (CosineCutoff), :60-73 (PolynomialCutoff)
//   spherical harmonics  e3nn 0.5.1 o3.SphericalHarmonics as built at nn/xpainn.py:49-51
//                        and called on vec[:, [1,2,0]] at nn/xpainn.py:71-74
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/xeq.h"

namespace xeq {

void set_error(const char* fmt, ...);
void note_launch(const char* name);   // counts and names the kernel launches of this library (xeq_launch_count / xeq_launch_names; name: a string literal)

#define XEQ_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      xeq::set_error(__VA_ARGS__);      \
      return XEQ_ERR_INVALID_ARGUMENT;  \
    }                                   \
  } while (0)

#define XEQ_CHECK_LAUNCH(name)                                                   \
  do {                                                                           \
    xeq::note_launch(name);                                                      \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      xeq::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return XEQ_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

// dtype dispatch: BODY sees `T`
#define XEQ_DISPATCH_FLOAT(dtype, ...)                         \
  do {                                                         \
    if ((dtype) == XEQ_F32) {                                  \
      using T = float;                                         \
      __VA_ARGS__                                              \
    } else if ((dtype) == XEQ_F64) {                           \
      using T = double;                                        \
      __VA_ARGS__                                              \
    } else {                                                   \
      xeq::set_error("unsupported dtype %d", (int)(dtype));    \
      return XEQ_ERR_INVALID_ARGUMENT;                         \
    }                                                          \
  } while (0)

// Irreps layout: blocks l = 0,1,2 in ascending order, `mul[l]` channels each
// (0 = absent), e3nn mul_ir layout (channel-major, m-minor).
struct Irreps {
  int mul[3];
  __host__ __device__ int C() const { return mul[0] + mul[1] + mul[2]; }
  __host__ __device__ int D() const { return mul[0] + 3 * mul[1] + 5 * mul[2]; }
  // channel u -> (l, flat offset of its first component)
  __host__ __device__ void locate(int u, int& l, int& off) const {
    if (u < mul[0]) {
      l = 0;
      off = u;
    } else if (u < mul[0] + mul[1]) {
      l = 1;
      off = mul[0] + 3 * (u - mul[0]);
    } else {
      l = 2;
      off = mul[0] + 3 * mul[1] + 5 * (u - mul[0] - mul[1]);
    }
  }
};

// Addressing of an equivariant row for channel u: elem(n, m) = off + n * node + m * comp.
//   layout 0 (e3nn mul_ir, what crosses the module boundary):  off = flat offset of u, node = D, comp = 1
//   layout 1 (BT, internal: block-major over l, then node, m, channel; see xeq_node.hip):
//            off = N * base_l + u', node = (2l+1) mul_l, comp = mul_l
struct XAddr {
  int64_t off, node;
  int comp;
};
__host__ __device__ inline XAddr xaddr(const Irreps& ir, int64_t N, int u, int layout) {
  int l, off;
  ir.locate(u, l, off);
  XAddr a;
  if (layout == 0) {
    a.off = off;
    a.node = ir.D();
    a.comp = 1;
  } else {
    const int up = u - (l == 0 ? 0 : (l == 1 ? ir.mul[0] : ir.mul[0] + ir.mul[1]));
    const int64_t base = l == 0 ? 0 : (l == 1 ? (int64_t)ir.mul[0] : (int64_t)ir.mul[0] + 3 * ir.mul[1]);
    a.off = N * base + up;
    a.node = (int64_t)(2 * l + 1) * ir.mul[l];
    a.comp = ir.mul[l];
  }
  return a;
}

// radial-basis / envelope selection (mirrors resolve_rbf / resolve_cutoff, nn/rbf.py:9-32)
struct RadialSpec {
  int rbf_kind;     // XEQ_RBF_BESSEL | XEQ_RBF_GAUSSIAN | XEQ_RBF_EXPBERN | XEQ_RBF_EXPNORM
  int cutoff_kind;  // XEQ_CUTOFF_COSINE | XEQ_CUTOFF_POLYNOMIAL
  int num_basis;
  double cutoff;
};

template <typename T> __device__ __forceinline__ void sincos_(T x, T* s, T* c);
template <> __device__ __forceinline__ void sincos_<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <> __device__ __forceinline__ void sincos_<double>(double x, double* s, double* c) { sincos(x, s, c); }
template <typename T> __device__ __forceinline__ T sqrt_(T x);
template <> __device__ __forceinline__ float sqrt_<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double sqrt_<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T exp_(T x);
template <> __device__ __forceinline__ float exp_<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double exp_<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T abs_(T x) { return x < T(0) ? -x : x; }
template <typename T> __device__ __forceinline__ T expm1_(T x);
template <> __device__ __forceinline__ float expm1_<float>(float x) { return expm1f(x); }
template <> __device__ __forceinline__ double expm1_<double>(double x) { return expm1(x); }
template <typename T> __device__ __forceinline__ T log_(T x);
template <> __device__ __forceinline__ float log_<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double log_<double>(double x) { return log(x); }

// envelope f(d) and f'(d)
template <typename T>
__device__ __forceinline__ void envelope(int kind, T d, T rc, T& f, T& df) {
  const T PI = T(3.14159265358979323846);
  if (!(d < rc)) {  // torch.where(dist < cutoff, ..., 0)  nn/rbf.py:47-48
    f = T(0);
    df = T(0);
    return;
  }
  if (kind == XEQ_CUTOFF_COSINE) {
    T s, c;
    sincos_<T>(PI * d / rc, &s, &c);
    f = T(0.5) * (c + T(1));
    df = -T(0.5) * PI / rc * s;
  } else {  // polynomial, order p = 3 (nn/rbf.py:60-73)
    const T p = T(3);
    T r = d / rc;
    T r2 = r * r, r3 = r2 * r, r4 = r3 * r, r5 = r4 * r;
    f = T(1) - T(0.5) * (p + 1) * (p + 2) * r3 + p * (p + 2) * r4 - T(0.5) * p * (p + 1) * r5;
    df = (-T(0.5) * (p + 1) * (p + 2) * p * r2 + p * (p + 2) * (p + 1) * r3 - T(0.5) * p * (p + 1) * (p + 2) * r4) / rc;
  }
}

// radial basis rho_k(d) and rho_k'(d); p0/p1 = freq/- (bessel), mean/std (gaussian), softplus(_alpha)/log binomial (exponential
// Bernstein: k and the basis size B select the polynomial), beta/mu (exponential norm)
template <typename T>
__device__ __forceinline__ void radial(int kind, T d, T rc, T p0, T p1, T& rho, T& drho, int k = 0, int B = 0) {
  if (kind == XEQ_RBF_EXPBERN) {
    // nn/rbf.py:186-191: x = -alpha d, rho_k = exp(logc_k + n_k x + v_k log(-expm1(x))), v_k = k, n_k = B - 1 - k
    const T x = -p0 * d;
    const T om = -expm1_<T>(x);           // 1 - e^x
    const T L = log_<T>(om);
    const T n = T(B - 1 - k), v = T(k);
    rho = exp_<T>(p1 + n * x + v * L);
    drho = rho * p0 * (v * (T(1) - om) / om - n);   // d/dd [n x + v log(1 - e^x)], x = -alpha d
  } else if (kind == XEQ_RBF_EXPNORM) {
    // nn/rbf.py:204-207: rho_k = exp(-beta_k (exp(-d) - mu_k)^2)
    const T e = exp_<T>(-d);
    const T t = e - p1;
    rho = exp_<T>(-p0 * t * t);
    drho = rho * T(2) * p0 * t * e;
  } else if (kind == XEQ_RBF_BESSEL) {
    const T eps = T(1e-5);
    T coeff = sqrt_<T>(T(2) / rc);
    T s, c;
    sincos_<T>(p0 * d, &s, &c);
    T inv = T(1) / (d + eps);
    rho = coeff * s * inv;
    drho = coeff * (p0 * c * inv - s * inv * inv);
  } else {  // gaussian smearing
    const T eps = T(1e-5);
    T sd = abs_<T>(p1) + eps;
    T coeff = T(1) / (sd * T(2.5066282746310002));  // sqrt(2 pi)
    T z = (d - p0) / sd;
    rho = coeff * exp_<T>(-T(0.5) * z * z);
    drho = -z / sd * rho;
  }
}

// d rho_k / d p0 and d rho_k / d p1 of the same bases: the trainable basis parameters (freq: nn/rbf.py:143-146; mean, std:
// nn/rbf.py:121-125) -- what a training pass contracts dL/d rho with (nn/training.py, xeq_message_param_grad)
template <typename T>
__device__ __forceinline__ void radial_dparam(int kind, T d, T rc, T p0, T p1, T& d0, T& d1) {
  const T eps = T(1e-5);
  if (kind == XEQ_RBF_BESSEL) {
    T coeff = sqrt_<T>(T(2) / rc);
    T s, c;
    sincos_<T>(p0 * d, &s, &c);
    d0 = coeff * d * c / (d + eps);
    d1 = T(0);
  } else {
    T sd = abs_<T>(p1) + eps;
    T coeff = T(1) / (sd * T(2.5066282746310002));
    T z = (d - p0) / sd;
    T rho = coeff * exp_<T>(-T(0.5) * z * z);
    d0 = rho * z / sd;
    // rho = exp(-z^2/2) / (sd sqrt(2 pi)), sd = |p1| + eps: d rho / d sd = rho (z^2 - 1) / sd; d|p1| / d p1 = sign (0 at 0, as autograd's abs)
    T sg = p1 > T(0) ? T(1) : (p1 < T(0) ? T(-1) : T(0));
    d1 = rho * (z * z - T(1)) / sd * sg;
  }
}

// Geometry of one edge.  rhat = r / max(|r|, 1e-12) (F.normalize, e3nn normalize=True).
template <typename T>
struct EdgeGeom {
  T d, inv_d;      // |r|, 1/max(|r|,1e-12)
  T x, y, z;       // unit vector in ORIGINAL axis order
};

template <typename T>
__device__ __forceinline__ EdgeGeom<T> edge_geom(T rx, T ry, T rz) {
  EdgeGeom<T> g;
  g.d = sqrt_<T>(rx * rx + ry * ry + rz * rz);
  T dn = g.d > T(1e-12) ? g.d : T(1e-12);
  g.inv_d = T(1) / dn;
  g.x = rx * g.inv_d;
  g.y = ry * g.inv_d;
  g.z = rz * g.inv_d;
  return g;
}

// Component-normalised SH of the unit vector, as the reference evaluates them:
// e3nn's (x,y,z) <- original (y,z,x).  Y[0]=1 implicit; y1[3], y2[5].
template <typename T>
__device__ __forceinline__ void sph_harm_l12(const EdgeGeom<T>& g, T* y1, T* y2) {
  const T S3 = T(1.7320508075688772), S5 = T(2.23606797749979), S15 = T(3.872983346207417);
  y1[0] = S3 * g.y;
  y1[1] = S3 * g.z;
  y1[2] = S3 * g.x;
  y2[0] = S15 * g.x * g.y;
  y2[1] = S15 * g.y * g.z;
  y2[2] = S5 * (g.z * g.z - T(0.5) * (g.x * g.x + g.y * g.y));
  y2[3] = S15 * g.x * g.z;
  y2[4] = T(0.5) * S15 * (g.x * g.x - g.y * g.y);
}

// Chain rule from (dL/dd, dL/dY1[3], dL/dY2[5]) to dL/dr (original axis order).
template <typename T>
__device__ __forceinline__ void edge_grad(const EdgeGeom<T>& g, T gd, const T* p1, const T* p2, T* out) {
  const T S3 = T(1.7320508075688772), S5 = T(2.23606797749979), S15 = T(3.872983346207417);
  // G = dL/d rhat
  T Gx = S3 * p1[2] + S15 * g.y * p2[0] - S5 * g.x * p2[2] + S15 * g.z * p2[3] + S15 * g.x * p2[4];
  T Gy = S3 * p1[0] + S15 * g.x * p2[0] + S15 * g.z * p2[1] - S5 * g.y * p2[2] - S15 * g.y * p2[4];
  T Gz = S3 * p1[1] + S15 * g.y * p2[1] + T(2) * S5 * g.z * p2[2] + S15 * g.x * p2[3];
  T Gr = Gx * g.x + Gy * g.y + Gz * g.z;
  bool ok = g.d > T(1e-12);  // F.normalize clamps the norm: no gradient through it below the clamp
  T s = ok ? g.inv_d : T(0);
  T gdd = ok ? gd : T(0);    // d|r|/dr = rhat (torch.linalg.norm backward is 0 at r = 0)
  out[0] = gdd * g.x + s * (Gx - Gr * g.x);
  out[1] = gdd * g.y + s * (Gy - Gr * g.y);
  out[2] = gdd * g.z + s * (Gz - Gr * g.z);
}

// ---- node tiles of the matrix-core node kernels (xeq_mlp.hip, xeq_update.hip) ---------------------------------------
// A launch costs (rounds of one workgroup per CU) x (time of a lone workgroup), so whole tiles run one workgroup each while
// they come in multiples of the CU count, and the remainder (or everything, for MD-sized systems) is SPLIT: `split` workgroups
// share a tile, each repeats its cheap first phase and takes every split-th group of output tiles / jobs.
// Workgroup b < n_full: tile b alone; else tile n_full + (b - n_full) / split, part (b - n_full) % split.
// Row count up to which the node-side products take their few-row forms (16 x 16 exact-f32 tiles: a quarter of the k-chain per wave,
// four times the waves; BIT-EQUAL results, see xeq_linear.hip).  XEQ_SMALL_ROWS overrides (0: never; read per call: tests flip it).
inline int64_t xeq_small_rows() {
  const char* v = getenv("XEQ_SMALL_ROWS");
  return v ? atoll(v) : 3584;   // 224 row tiles: measured crossover of a whole replayed step between 3.6 k and 4.6 k atoms (profiles/r06_small_experiments.txt 13)
}

struct TileSplit {
  int n_full, split;
  __host__ __device__ unsigned grid(int64_t tiles) const { return (unsigned)(n_full + (tiles - n_full) * split); }
  __device__ __forceinline__ void decode(int b, int& tile, int& part, int& parts) const {
    if (b < n_full) {
      tile = b;
      part = 0;
      parts = 1;
    } else {
      const int r = b - n_full;
      tile = n_full + r / split;
      part = r - (tile - n_full) * split;
      parts = split;
    }
  }
};
// max_split: the number of pieces the second phase can be cut into (>= 1); 256 CUs (MI355X)
inline TileSplit tile_split(int64_t tiles, int max_split) {
  TileSplit t{(int)tiles, 1};
  if (max_split < 2 || tiles <= 0) return t;
  const int64_t cus = 256, rem = tiles % cus;
  if (tiles * 2 <= cus) {                 // few tiles: split them all
    t.n_full = 0;
    t.split = (int)(cus / tiles < max_split ? cus / tiles : max_split);
  } else if (rem > 0 && rem * 2 <= cus) { // a short last round: split its tiles over the CUs it would leave idle
    t.n_full = (int)(tiles - rem);
    t.split = (int)(cus / rem < max_split ? cus / rem : max_split);
  }
  if (t.split < 1) t.split = 1;
  return t;
}

// ---- wave64 helpers ---------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Exclusive prefix sums of f(0 .. n - 1) by ONE workgroup of SCAN_WG_THREADS threads in one launch (the library's grid-wide scan is
// two launches; a captured step pays ~5 us per launch, and the arrays scanned on the path -- degrees, quads per node -- are a few
// 10 k entries).  The values are staged in LDS with coalesced loads (a first form had every thread walk its own contiguous run in
// global memory: 64 cache lines per wave instruction, 21 us for 18 k entries), thread t then owns the run [t c, (t + 1) c) of the LDS
// copy (c odd: conflict-free banks), the run totals are scanned over the workgroup and the runs rewritten in place as exclusive sums.
// Returns the TOTAL to every thread; vals[0 .. n) then hold the prefix sums and the caller copies them out coalesced (so it can
// guard the result by the total first).  LDS: (n + SCAN_WG_THREADS / 64 + 1) ints (dynamic).
constexpr int SCAN_WG_THREADS = 1024;
constexpr int64_t SCAN_WG_MAX_ITEMS = 36 * 1024;   // 144 KB of LDS; above: the grid-wide scan
template <typename F>
__device__ __forceinline__ int32_t wg_scan_lds(const F& f, int64_t n, int32_t* vals) {
  int32_t* aux = vals + n;   // [SCAN_WG_THREADS / 64 + 1]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll 4
  for (int64_t i = t; i < n; i += SCAN_WG_THREADS) vals[i] = f(i);
  __syncthreads();
  const int64_t c = ((n + SCAN_WG_THREADS - 1) / SCAN_WG_THREADS) | 1;
  const int64_t i0 = min((int64_t)t * c, n), i1 = min(i0 + c, n);
  int32_t sum = 0;
  for (int64_t i = i0; i < i1; ++i) sum += vals[i];
  int32_t incl = sum;   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) aux[wave] = incl;
  __syncthreads();
  if (t == 0) {
    int32_t run = 0;
    for (int w = 0; w < SCAN_WG_THREADS / 64; ++w) {
      const int32_t v = aux[w];
      aux[w] = run;
      run += v;
    }
    aux[SCAN_WG_THREADS / 64] = run;
  }
  __syncthreads();
  int32_t base = aux[wave] + incl - sum;
  for (int64_t i = i0; i < i1; ++i) {
    const int32_t v = vals[i];
    vals[i] = base;
    base += v;
  }
  const int32_t total = aux[SCAN_WG_THREADS / 64];
  __syncthreads();
  return total;
}
inline size_t wg_scan_lds_bytes(int64_t n) { return sizeof(int32_t) * (size_t)(n + SCAN_WG_THREADS / 64 + 1); }

// XCD-aware persistent work mapping (speed only; every item is visited exactly
// once for ANY placement).  Items are cut in chunks of `chunk` consecutive items;
// chunk q belongs to label q % 8; the blocks with blockIdx % 8 == label walk that
// label's items.  Blocks b and b+8 share an XCD under round-robin dispatch
// (MI355X_MICROARCH.md, Workgroup dispatch), so consecutive nodes (one molecule)
// are served by one XCD's L2.
struct XcdWalk {
  int64_t n_items;
  int chunk;
  int label, slot, n_slots;
  int64_t idx;
  __device__ XcdWalk(int64_t n, int chunk_) : n_items(n), chunk(chunk_) {
    int nb = gridDim.x;
    if (nb >= 8) {
      label = blockIdx.x & 7;
      slot = blockIdx.x >> 3;
      n_slots = (nb - label + 7) >> 3;  // blocks with this label
    } else {  // tiny grid: no labelling
      label = -1;
      slot = blockIdx.x;
      n_slots = nb;
    }
    idx = slot;
  }
  // returns the next item or -1
  __device__ int64_t next() {
    while (true) {
      int64_t item;
      if (label < 0) {
        item = idx;
        if (item >= n_items) return -1;
      } else {
        int64_t q8 = idx / chunk;
        int r = (int)(idx - q8 * chunk);
        int64_t q = q8 * 8 + label;
        item = q * chunk + r;
        if (q * chunk >= n_items) return -1;
        if (item >= n_items) {  // ragged last chunk
          idx += n_slots;
          continue;
        }
      }
      idx += n_slots;
      return item;
    }
  }
};

}  // namespace xeq
