// Operator-level kernels: edge geometry, e3nn-style elementwise ops, equivariant
// layer norm, segmented sums.
#include "xeq_common.h"

namespace xeq {

// ------------------------------------------------------------- edge vectors
template <typename T>
__global__ void k_edge_vectors_fwd(const T* __restrict__ pos, const int64_t* __restrict__ edge_index, int64_t E,
                                   const T* __restrict__ cell, const T* __restrict__ cell_offsets,
                                   const int64_t* __restrict__ batch, T* __restrict__ vec, T* __restrict__ dist) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t c = edge_index[e], n = edge_index[E + e];
  T v[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) v[a] = pos[3 * c + a] - pos[3 * n + a];
  if (cell != nullptr) {
    const T* cm = cell + (batch ? 9 * batch[n] : 0);  // cell of the NEIGHBOR's graph (basic.py:125-126)
    T o0 = cell_offsets[3 * e], o1 = cell_offsets[3 * e + 1], o2 = cell_offsets[3 * e + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] -= o0 * cm[a] + o1 * cm[3 + a] + o2 * cm[6 + a];  // einsum ni,nij->nj
  }
  vec[3 * e] = v[0];
  vec[3 * e + 1] = v[1];
  vec[3 * e + 2] = v[2];
  if (dist) dist[e] = sqrt_<T>(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

// grad_pos[i] = sum_{center=i} g[e] - sum_{neighbor=i} g[e].  16 lanes per node: lane l takes entries l, l+16, ... of the
// node's two CSR segments (all three axes), then a fixed xor-butterfly over the 16 lanes: deterministic, and a node
// with 50-100 edges (periodic boxes) costs 4-7 rounds of independent loads instead of a serial walk
template <typename T>
__global__ void __launch_bounds__(256) k_edge_vectors_bwd(const T* __restrict__ g, int64_t N, const int32_t* __restrict__ c_rowptr,
                                   const int32_t* __restrict__ c_perm, const int32_t* __restrict__ n_rowptr,
                                   const int32_t* __restrict__ n_perm, T* __restrict__ grad_pos) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t >> 4;
  const int l = (int)(t & 15);
  T a0 = T(0), a1 = T(0), a2 = T(0);
  if (i < N) {
    for (int32_t p = c_rowptr[i] + l, p1 = c_rowptr[i + 1]; p < p1; p += 16) {
      const int64_t e = c_perm ? c_perm[p] : p;
      a0 += g[3 * e];
      a1 += g[3 * e + 1];
      a2 += g[3 * e + 2];
    }
    for (int32_t p = n_rowptr[i] + l, p1 = n_rowptr[i + 1]; p < p1; p += 16) {
      const int64_t e = n_perm ? n_perm[p] : p;
      if (e < 0) continue;   // a mirror map's missing entry (xeq_reverse_edge_map_pbc)
      a0 -= g[3 * e];
      a1 -= g[3 * e + 1];
      a2 -= g[3 * e + 2];
    }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    a0 += __shfl_xor(a0, m, 16);
    a1 += __shfl_xor(a1, m, 16);
    a2 += __shfl_xor(a2, m, 16);
  }
  if (i < N && l == 0) {
    grad_pos[3 * i] = a0;
    grad_pos[3 * i + 1] = a1;
    grad_pos[3 * i + 2] = a2;
  }
}

// ------------------------------------------------------- spherical harmonics
// `vec` arrives in e3nn axis order (x_e, y_e, z_e) = original (y, z, x).
template <typename T>
__global__ void k_sph_harm_fwd(const T* __restrict__ vec, int64_t n, Irreps ir, int normalize,
                               T* __restrict__ out) {
  const int D = ir.D();
  int64_t e = blockIdx.x;
  if (e >= n) return;
  T xe = vec[3 * e], ye = vec[3 * e + 1], ze = vec[3 * e + 2];
  // back to original order: x = z_e, y = x_e, z = y_e
  EdgeGeom<T> g;
  if (normalize) {
    g = edge_geom<T>(ze, xe, ye);
  } else {
    g.x = ze; g.y = xe; g.z = ye; g.d = T(1); g.inv_d = T(1);
  }
  T y1[3], y2[5];
  sph_harm_l12<T>(g, y1, y2);
  for (int f = threadIdx.x; f < D; f += blockDim.x) {
    T v;
    if (f < ir.mul[0]) v = T(1);
    else if (f < ir.mul[0] + 3 * ir.mul[1]) v = y1[(f - ir.mul[0]) % 3];
    else v = y2[(f - ir.mul[0] - 3 * ir.mul[1]) % 5];
    out[e * D + f] = v;
  }
}

template <typename T>
__global__ void k_sph_harm_bwd(const T* __restrict__ vec, const T* __restrict__ grad_out, int64_t n, Irreps ir,
                               int normalize, T* __restrict__ grad_vec) {
  // one wave per edge: reduce grad_out over the repeated copies, then chain rule
  const int D = ir.D();
  const int lane = threadIdx.x & 63;
  int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= n) return;
  T p1[3] = {T(0), T(0), T(0)}, p2[5] = {T(0), T(0), T(0), T(0), T(0)};
  const int o1 = ir.mul[0], o2 = ir.mul[0] + 3 * ir.mul[1];
  for (int f = o1 + lane; f < D; f += 64) {
    T gv = grad_out[e * D + f];
    if (f < o2) {
      int m = (f - o1) % 3;
#pragma unroll
      for (int q = 0; q < 3; ++q) p1[q] += (m == q) ? gv : T(0);
    } else {
      int m = (f - o2) % 5;
#pragma unroll
      for (int q = 0; q < 5; ++q) p2[q] += (m == q) ? gv : T(0);
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) p1[q] = wave_sum<T>(p1[q]);
#pragma unroll
  for (int q = 0; q < 5; ++q) p2[q] = wave_sum<T>(p2[q]);
  if (lane == 0) {
    T xe = vec[3 * e], ye = vec[3 * e + 1], ze = vec[3 * e + 2];
    T out[3];
    if (normalize) {
      EdgeGeom<T> g = edge_geom<T>(ze, xe, ye);
      edge_grad<T>(g, T(0), p1, p2, out);
    } else {
      // polynomial in the raw vector: reuse edge_grad's dY/drhat with projection disabled
      EdgeGeom<T> g;
      g.x = ze; g.y = xe; g.z = ye; g.d = T(1); g.inv_d = T(1);
      const T S3 = T(1.7320508075688772), S5 = T(2.23606797749979), S15 = T(3.872983346207417);
      out[0] = S3 * p1[2] + S15 * g.y * p2[0] - S5 * g.x * p2[2] + S15 * g.z * p2[3] + S15 * g.x * p2[4];
      out[1] = S3 * p1[0] + S15 * g.x * p2[0] + S15 * g.z * p2[1] - S5 * g.y * p2[2] - S15 * g.y * p2[4];
      out[2] = S3 * p1[1] + S15 * g.y * p2[1] + T(2) * S5 * g.z * p2[2] + S15 * g.x * p2[3];
    }
    // out is d/d(x,y,z) original; grad_vec in e3nn order (x_e,y_e,z_e) = (y,z,x)
    grad_vec[3 * e] = out[1];
    grad_vec[3 * e + 1] = out[2];
    grad_vec[3 * e + 2] = out[0];
  }
}

// ------------------------------------------------------------------- radial
template <typename T>
__global__ void k_radial_fwd(const T* __restrict__ dist, int64_t n, RadialSpec rs, const T* __restrict__ p0,
                             const T* __restrict__ p1, T* __restrict__ rbf, T* __restrict__ fcut) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int B = rs.num_basis;
  if (t >= n * (B + 1)) return;
  int64_t e = t / (B + 1);
  int k = (int)(t - e * (B + 1));
  T d = dist[e];
  T rc = (T)rs.cutoff;
  if (k == B) {
    if (fcut) {
      T f, df;
      envelope<T>(rs.cutoff_kind, d, rc, f, df);
      fcut[e] = f;
    }
  } else if (rbf) {
    T rho, drho;
    radial<T>(rs.rbf_kind, d, rc, p0[k], p1 ? p1[k] : T(0), rho, drho, k, B);
    rbf[e * B + k] = rho;
  }
}

// ------------------------------------------------ elementwise TP / channel dot
template <typename T>
__global__ void k_elementwise_tp(const T* __restrict__ x, const T* __restrict__ g, int64_t n, int64_t g_rows,
                                 Irreps ir, T* __restrict__ out) {
  const int D = ir.D(), C = ir.C();
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * D) return;
  int64_t row = t / D;
  int f = (int)(t - row * D);
  int u;
  if (f < ir.mul[0]) u = f;
  else if (f < ir.mul[0] + 3 * ir.mul[1]) u = ir.mul[0] + (f - ir.mul[0]) / 3;
  else u = ir.mul[0] + ir.mul[1] + (f - ir.mul[0] - 3 * ir.mul[1]) / 5;
  int64_t grow = g_rows == 1 ? 0 : row;
  out[t] = x[t] * g[grow * C + u];
}

template <typename T>
__global__ void k_channel_dot(const T* __restrict__ a, const T* __restrict__ b, int64_t n, Irreps ir,
                              T* __restrict__ out) {
  const int D = ir.D(), C = ir.C();
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * C) return;
  int64_t row = t / C;
  int u = (int)(t - row * C);
  int l, off;
  ir.locate(u, l, off);
  T acc = T(0);
  for (int m = 0; m < 2 * l + 1; ++m) acc += a[row * D + off + m] * b[row * D + off + m];
  out[t] = acc;
}

// ------------------------------------------------------ equivariant layer norm
// One wave per node.  y = (x - [l=0] mean_0e(x)) * rsqrt(mean_u sum_m xc^2 + eps) * w_u + [l=0] b_u
template <typename T>
__global__ void k_eqln_fwd(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ bias,
                           int64_t n, Irreps ir, T eps, T* __restrict__ out) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  const T* xr = x + row * D;
  T s = T(0);
  for (int f = lane; f < m0; f += 64) s += xr[f];
  T mean = m0 > 0 ? wave_sum<T>(s) / T(m0) : T(0);
  T sq = T(0);
  for (int f = lane; f < D; f += 64) {
    T v = xr[f] - (f < m0 ? mean : T(0));
    sq += v * v;
  }
  sq = wave_sum<T>(sq);
  T r = T(1) / sqrt_<T>(sq / T(C) + eps);
  for (int f = lane; f < D; f += 64) {
    int u;
    if (f < m0) u = f;
    else if (f < m0 + 3 * ir.mul[1]) u = m0 + (f - m0) / 3;
    else u = m0 + ir.mul[1] + (f - m0 - 3 * ir.mul[1]) / 5;
    T v = (xr[f] - (f < m0 ? mean : T(0))) * r * w[u];
    if (f < m0) v += bias[f];
    out[row * D + f] = v;
  }
}

template <typename T>
__global__ void k_eqln_bwd(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ go, int64_t n,
                           Irreps ir, T eps, T* __restrict__ gx) {
  const int D = ir.D(), C = ir.C(), m0 = ir.mul[0];
  const int lane = threadIdx.x & 63;
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  const T* xr = x + row * D;
  const T* gr = go + row * D;
  T s = T(0);
  for (int f = lane; f < m0; f += 64) s += xr[f];
  T mean = m0 > 0 ? wave_sum<T>(s) / T(m0) : T(0);
  T sq = T(0), dotp = T(0);
  for (int f = lane; f < D; f += 64) {
    int u;
    if (f < m0) u = f;
    else if (f < m0 + 3 * ir.mul[1]) u = m0 + (f - m0) / 3;
    else u = m0 + ir.mul[1] + (f - m0 - 3 * ir.mul[1]) / 5;
    T xc = xr[f] - (f < m0 ? mean : T(0));
    sq += xc * xc;
    dotp += gr[f] * w[u] * xc;  // sum dy * xc
  }
  sq = wave_sum<T>(sq);
  dotp = wave_sum<T>(dotp);
  T r = T(1) / sqrt_<T>(sq / T(C) + eps);
  T coef = dotp * r * r * r / T(C);
  // gxc = r * dy - coef * xc ; then remove the mean of gxc over the 0e channels
  T gs = T(0);
  for (int f = lane; f < m0; f += 64) {
    T xc = xr[f] - mean;
    gs += r * gr[f] * w[f] - coef * xc;
  }
  T gmean = m0 > 0 ? wave_sum<T>(gs) / T(m0) : T(0);
  for (int f = lane; f < D; f += 64) {
    int u;
    if (f < m0) u = f;
    else if (f < m0 + 3 * ir.mul[1]) u = m0 + (f - m0) / 3;
    else u = m0 + ir.mul[1] + (f - m0 - 3 * ir.mul[1]) / 5;
    T xc = xr[f] - (f < m0 ? mean : T(0));
    T v = r * gr[f] * w[u] - coef * xc;
    if (f < m0) v -= gmean;
    gx[row * D + f] = v;
  }
}

// ------------------------------------------------------------- segmented sums
// one wave per (segment, 64-column slab); rows summed in order => reproducible
template <typename T>
__global__ void k_segment_sum(const T* __restrict__ src, const int64_t* __restrict__ ptr, int64_t n_seg,
                              int64_t width, T* __restrict__ out) {
  int64_t slabs = (width + 63) / 64;
  int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= n_seg * slabs) return;
  int64_t g = wid / slabs;
  int64_t col = (wid - g * slabs) * 64 + lane;
  int64_t a = ptr[g], b = ptr[g + 1];
  if (width == 1) {  // scalar rows: lanes stride the rows, then a wave reduction
    T acc = T(0);
    for (int64_t i = a + lane; i < b; i += 64) acc += src[i];
    acc = wave_sum<T>(acc);
    if (lane == 0) out[g] = acc;
    return;
  }
  if (col >= width) return;
  T acc = T(0);
  for (int64_t i = a; i < b; ++i) acc += src[i * width + col];
  out[g * width + col] = acc;
}

template <typename T>
__global__ void k_scatter_add(const T* __restrict__ src, const int64_t* __restrict__ index, int64_t n,
                              int64_t width, T* __restrict__ out, int64_t n_out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  int64_t row = t / width;
  int64_t col = t - row * width;
  int64_t dst = index[row];
  if (dst < 0 || dst >= n_out) return;
  atomicAdd(&out[dst * width + col], src[t]);
}

static inline int check_irreps(const int32_t mul[3], Irreps& ir, const char* who) {
  for (int l = 0; l < 3; ++l) {
    if (mul[l] < 0) {
      set_error("%s: negative multiplicity", who);
      return XEQ_ERR_INVALID_ARGUMENT;
    }
    ir.mul[l] = mul[l];
  }
  if (ir.C() == 0) {
    set_error("%s: empty irreps", who);
    return XEQ_ERR_INVALID_ARGUMENT;
  }
  return XEQ_OK;
}

// several small device-to-device copies in one launch (blockIdx.y = buffer); HIP-graph replay refreshes its captured
// inputs with it instead of a dozen 3 us copy launches
struct CopyMany {
  const char* src[XEQ_COPY_MANY_MAX];
  char* dst[XEQ_COPY_MANY_MAX];
  int64_t bytes[XEQ_COPY_MANY_MAX];
  bool wide[XEQ_COPY_MANY_MAX];
};
__global__ void k_copy_many(CopyMany cm) {
  const int b = blockIdx.y;
  const int64_t bytes = cm.bytes[b];
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  if (cm.wide[b]) {
    const uint4* s = reinterpret_cast<const uint4*>(cm.src[b]);
    uint4* d = reinterpret_cast<uint4*>(cm.dst[b]);
    for (int64_t i = i0; i < bytes / 16; i += stride) d[i] = s[i];
  } else {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(cm.src[b]);
    uint32_t* d = reinterpret_cast<uint32_t*>(cm.dst[b]);
    for (int64_t i = i0; i < bytes / 4; i += stride) d[i] = s[i];
  }
}

// several pairs of small device buffers compared in one launch: flag[0] = gen when any 4-byte word differs (the caller hands a NEW gen
// every call, so the flag needs no clearing launch)
__global__ void k_compare_many(CopyMany cm, int32_t gen, int32_t* __restrict__ flag) {
  const int b = blockIdx.y;
  const int64_t words = cm.bytes[b] / 4;
  const uint32_t* x = reinterpret_cast<const uint32_t*>(cm.src[b]);
  const uint32_t* y = reinterpret_cast<const uint32_t*>(cm.dst[b]);
  bool diff = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) diff |= x[i] != y[i];
  if (__ballot(diff) != 0ull && (threadIdx.x & 63) == 0) flag[0] = gen;   // (every writer writes the same value)
}

}  // namespace xeq

using namespace xeq;

#define XEQ_IRREPS(who)                       \
  Irreps ir;                                  \
  {                                           \
    int rc_ = check_irreps(mul, ir, who);     \
    if (rc_ != XEQ_OK) return rc_;            \
  }

#include <atomic>
static std::atomic<long long> g_pack_epoch{0};

extern "C" {

/* Epoch of the packed weight copies (fragment-order copies of nn.Linear / o3.Linear weights that the matrix-core kernels read).  Every
 * pack cache -- nn/fused.py, nn/nodeblock.py, csrc/xeq_torch.cpp -- keys on it next to the tensors' version counters: whatever changes
 * parameters behind autograd's back (a captured optimizer step replayed as a graph, train.GraphedTrainStep) bumps it, and the next
 * evaluation repacks. */
long long xeq_pack_epoch(void) { return g_pack_epoch.load(); }
void xeq_pack_epoch_bump(void) { g_pack_epoch.fetch_add(1); }


/* ---- launch policy shared by every front (Python modules, registered operator): ONE statement of the rules ---------------- */
int xeq_message_auto_family(int dtype, int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]) {
  // tiny graphs are launch-bound and the scalar-broadcast kernels need no walk plan; everything else that fits takes the wave / quad
  // matrix-core kernels (since the split-bf16 filter also the dense periodic boxes; num_basis <= 31), then sb (f64, other channel
  // layouts), then the generic 64-bit form
  const bool f32 = dtype == XEQ_F32;
  const bool sb_fits = xeq_message_sb_fits(n_nodes, n_edges, num_basis, node_dim, mul) != 0;
  if (n_edges < 4096 && sb_fits) return XEQ_FAMILY_SB;
  if (f32 && xeq_message_wq_fits(n_nodes, n_edges, num_basis, node_dim, mul)) return XEQ_FAMILY_WQ;
  if (sb_fits) return XEQ_FAMILY_SB;
  return XEQ_FAMILY_GENERIC;
}

/* rows up to which the node-side products take their few-row forms (16 x 16 exact-f32 tiles, bit-equal to the 32-row forms;
 * csrc/xeq_linear_s.h): 3 584 unless XEQ_SMALL_ROWS says otherwise (0: never).  Host only. */
int64_t xeq_small_rows_limit(void) { return xeq_small_rows(); }

int xeq_message_wq_edges_per_stream(int64_t n_nodes, int64_t n_edges) {
  // a step (eight half-wave streams) should gather from few enough nodes for its window to fit LDS: 64 edges per stream is ~30 owner
  // nodes; small systems get shorter streams so that the launch still spreads over the chip, never below the mean segment length.
  // The cap was 64 up to round 4; since the workgroups stage their unit's weights from the packed copy (round 5) longer steps pay:
  // QM9-1024 2.287 / 2.261 / 2.270 / 2.265 ms at 64 / 72 / 80 / 88, MD17 x 4096 9.20 / 9.07 / 8.92 / 8.99 ms, QM9 x 8192 14.96 / 14.90 /
  // 14.76 / 14.61 ms (one box): 80.
  const double per_node = (double)n_edges / (double)(n_nodes > 0 ? n_nodes : 1);
  double v = per_node > 16.0 ? per_node : 16.0;
  const double spread = (double)n_edges / 1500.0;
  if (spread > v) v = spread;
  // (Round 6 tried 48-edge table streams walked three at a time by the l = 0 units -- two stream classes,
  // csrc/xeq_message_wq.hip::wq_long_mult: slower with all units in one launch, 1.731 -> 1.775 ms; the one length of 80 stays.)
  if (v > 80.0) v = 80.0;
  return (int)v;
}


int xeq_edge_vectors_fwd(int dtype, const void* pos, const int64_t* edge_index, int64_t n_edges,
                         const void* cell, const void* cell_offsets, const int64_t* batch, void* vec, void* dist,
                         void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0, "xeq_edge_vectors_fwd: n_edges < 0");
  XEQ_CHECK_ARG((cell == nullptr) == (cell_offsets == nullptr), "xeq_edge_vectors_fwd: cell and cell_offsets must both be given or both be NULL (data/transform.py:67-68)");
  if (n_edges == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_edge_vectors_fwd<T>), dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos, edge_index, n_edges, (const T*)cell,
                       (const T*)cell_offsets, batch, (T*)vec, (T*)dist);
  });
  XEQ_CHECK_LAUNCH("xeq_edge_vectors_fwd");
  return XEQ_OK;
}

int xeq_edge_vectors_bwd(int dtype, const void* grad_vec, int64_t n_nodes, const int32_t* c_rowptr,
                         const int32_t* c_perm, const int32_t* n_rowptr, const int32_t* n_perm, void* grad_pos,
                         void* stream) {
  XEQ_CHECK_ARG(n_nodes >= 0, "xeq_edge_vectors_bwd: n_nodes < 0");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_edge_vectors_bwd<T>), dim3((unsigned)((16 * n_nodes + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)grad_vec, n_nodes, c_rowptr, c_perm, n_rowptr, n_perm,
                       (T*)grad_pos);
  });
  XEQ_CHECK_LAUNCH("xeq_edge_vectors_bwd");
  return XEQ_OK;
}

int xeq_sph_harm_fwd(int dtype, const void* vec, int64_t n, const int32_t mul[3], int normalize, void* out,
                     void* stream) {
  XEQ_IRREPS("xeq_sph_harm_fwd");
  if (n <= 0) return XEQ_OK;
  XEQ_CHECK_ARG(n < (1ll << 31), "xeq_sph_harm_fwd: too many rows");
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_sph_harm_fwd<T>), dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream, (const T*)vec, n,
                       ir, normalize, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_sph_harm_fwd");
  return XEQ_OK;
}

int xeq_sph_harm_bwd(int dtype, const void* vec, const void* grad_out, int64_t n, const int32_t mul[3],
                     int normalize, void* grad_vec, void* stream) {
  XEQ_IRREPS("xeq_sph_harm_bwd");
  if (n <= 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_sph_harm_bwd<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)vec, (const T*)grad_out, n, ir, normalize, (T*)grad_vec);
  });
  XEQ_CHECK_LAUNCH("xeq_sph_harm_bwd");
  return XEQ_OK;
}

int xeq_radial_fwd(int dtype, const void* dist, int64_t n, int rbf_kind, int cutoff_kind, int num_basis,
                   double cutoff, const void* p0, const void* p1, void* rbf_out, void* fcut_out, void* stream) {
  XEQ_CHECK_ARG(num_basis > 0 && cutoff > 0, "xeq_radial_fwd: bad num_basis/cutoff");
  XEQ_CHECK_ARG(rbf_kind >= XEQ_RBF_BESSEL && rbf_kind <= XEQ_RBF_EXPNORM, "xeq_radial_fwd: rbf kernel %d is not implemented", rbf_kind);
  XEQ_CHECK_ARG(cutoff_kind == XEQ_CUTOFF_COSINE || cutoff_kind == XEQ_CUTOFF_POLYNOMIAL, "xeq_radial_fwd: cutoff function %d is not implemented", cutoff_kind);
  XEQ_CHECK_ARG(rbf_out == nullptr || p0 != nullptr, "xeq_radial_fwd: rbf parameters missing");
  XEQ_CHECK_ARG(rbf_out == nullptr || rbf_kind == XEQ_RBF_BESSEL || p1 != nullptr, "xeq_radial_fwd: this radial basis needs its second parameter array (std / logc / mu)");
  if (n <= 0) return XEQ_OK;
  RadialSpec rs{rbf_kind, cutoff_kind, num_basis, cutoff};
  int64_t total = n * (num_basis + 1);
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radial_fwd<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dist, n, rs, (const T*)p0, (const T*)p1, (T*)rbf_out, (T*)fcut_out);
  });
  XEQ_CHECK_LAUNCH("xeq_radial_fwd");
  return XEQ_OK;
}

int xeq_elementwise_tp_fwd(int dtype, const void* x, const void* g, int64_t n, int64_t g_rows,
                           const int32_t mul[3], void* out, void* stream) {
  XEQ_IRREPS("xeq_elementwise_tp_fwd");
  XEQ_CHECK_ARG(g_rows == 1 || g_rows == n, "xeq_elementwise_tp_fwd: gate rows %lld do not match %lld", (long long)g_rows, (long long)n);
  if (n <= 0) return XEQ_OK;
  int64_t total = n * ir.D();
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_elementwise_tp<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)x, (const T*)g, n, g_rows, ir, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_elementwise_tp_fwd");
  return XEQ_OK;
}

int xeq_channel_dot_fwd(int dtype, const void* a, const void* b, int64_t n, const int32_t mul[3], void* out,
                        void* stream) {
  XEQ_IRREPS("xeq_channel_dot_fwd");
  if (n <= 0) return XEQ_OK;
  int64_t total = n * ir.C();
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_channel_dot<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)a, (const T*)b, n, ir, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_channel_dot_fwd");
  return XEQ_OK;
}

int xeq_eqln_fwd(int dtype, const void* x, const void* weight, const void* bias, int64_t n,
                 const int32_t mul[3], double eps, void* out, void* stream) {
  XEQ_IRREPS("xeq_eqln_fwd");
  if (n <= 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_eqln_fwd<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)weight, (const T*)bias, n, ir, (T)eps, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_eqln_fwd");
  return XEQ_OK;
}

int xeq_eqln_bwd(int dtype, const void* x, const void* weight, const void* grad_out, int64_t n,
                 const int32_t mul[3], double eps, void* grad_x, void* stream) {
  XEQ_IRREPS("xeq_eqln_bwd");
  if (n <= 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_eqln_bwd<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (const T*)weight, (const T*)grad_out, n, ir, (T)eps, (T*)grad_x);
  });
  XEQ_CHECK_LAUNCH("xeq_eqln_bwd");
  return XEQ_OK;
}

int xeq_segment_sum(int dtype, const void* src, const int64_t* ptr, int64_t n_segments, int64_t width,
                    void* out, void* stream) {
  XEQ_CHECK_ARG(n_segments >= 0 && width > 0, "xeq_segment_sum: bad sizes");
  if (n_segments == 0) return XEQ_OK;
  int64_t waves = n_segments * ((width + 63) / 64);
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_segment_sum<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)src, ptr, n_segments, width, (T*)out);
  });
  XEQ_CHECK_LAUNCH("xeq_segment_sum");
  return XEQ_OK;
}

int xeq_scatter_add(int dtype, const void* src, const int64_t* index, int64_t n, int64_t width, void* out,
                    int64_t n_out, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && width > 0 && n_out >= 0, "xeq_scatter_add: bad sizes");
  if (n == 0) return XEQ_OK;
  int64_t total = n * width;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_scatter_add<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)src, index, n, width, (T*)out, n_out);
  });
  XEQ_CHECK_LAUNCH("xeq_scatter_add");
  return XEQ_OK;
}

/* development-free plumbing: see include/xeq.h */
int xeq_copy_many(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n <= XEQ_COPY_MANY_MAX, "xeq_copy_many: %d buffers (at most %d)", n, XEQ_COPY_MANY_MAX);
  xeq::CopyMany cm{};
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    XEQ_CHECK_ARG(bytes[i] >= 0 && bytes[i] % 4 == 0 && ((uintptr_t)src[i] % 4 == 0) && ((uintptr_t)dst[i] % 4 == 0),
                  "xeq_copy_many: buffer %d is not a whole number of aligned 4-byte words", i);
    cm.src[i] = (const char*)src[i];
    cm.dst[i] = (char*)dst[i];
    cm.bytes[i] = bytes[i];
    cm.wide[i] = bytes[i] % 16 == 0 && (uintptr_t)src[i] % 16 == 0 && (uintptr_t)dst[i] % 16 == 0;
    most = bytes[i] > most ? bytes[i] : most;
  }
  if (n == 0 || most == 0) return XEQ_OK;
  // 16 bytes per thread and trip; the widest buffer takes at most 64 workgroups per trip of the grid-stride loop
  const int64_t wg = (most / 16 + 255) / 256;
  hipLaunchKernelGGL(xeq::k_copy_many, dim3((unsigned)(wg < 1 ? 1 : (wg > 1024 ? 1024 : wg)), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, cm);
  XEQ_CHECK_LAUNCH("xeq_copy_many");
  return XEQ_OK;
}

/* n <= XEQ_COPY_MANY_MAX pairs of device buffers (whole 4-byte words) compared in ONE launch: flag[0] = gen if any pair differs, else
 * flag[0] is left alone -- hand a gen the flag has never held (a counter): no clearing launch, one read-back. */
int xeq_compare_many(int n, const void* const* a, const void* const* b, const int64_t* bytes, int32_t gen, int32_t* flag, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n <= XEQ_COPY_MANY_MAX && flag, "xeq_compare_many: %d buffers (at most %d) / null flag", n, XEQ_COPY_MANY_MAX);
  CopyMany cm;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    XEQ_CHECK_ARG(bytes[i] >= 0 && bytes[i] % 4 == 0 && ((uintptr_t)a[i] % 4 == 0) && ((uintptr_t)b[i] % 4 == 0),
                  "xeq_compare_many: buffer %d is not a whole number of aligned 4-byte words", i);
    cm.src[i] = (const char*)a[i];
    cm.dst[i] = (char*)const_cast<void*>(b[i]);
    cm.bytes[i] = bytes[i];
    cm.wide[i] = false;
    most = bytes[i] > most ? bytes[i] : most;
  }
  if (n == 0 || most == 0) return XEQ_OK;
  const int64_t wg = (most / 4 + 1023) / 1024;
  hipLaunchKernelGGL(xeq::k_compare_many, dim3((unsigned)(wg < 1 ? 1 : (wg > 1024 ? 1024 : wg)), (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     cm, gen, flag);
  XEQ_CHECK_LAUNCH("xeq_compare_many");
  return XEQ_OK;
}

}  // extern "C"

// ---- a batch into capacity-sized arrays, padded (xeq_load_padded_batch, include/xeq.h) ----
namespace xeq {
template <typename T, typename Z>
__global__ void k_load_padded_batch(const T* __restrict__ pos, const Z* __restrict__ z, const int64_t* __restrict__ ptr,
                                    const int64_t* __restrict__ batch, int64_t n, int64_t g, int64_t N, int64_t G, T pad0, T spacing,
                                    T* __restrict__ pos_out, int32_t* __restrict__ z_out, int64_t* __restrict__ ptr_out,
                                    int64_t* __restrict__ batch_out, int64_t atom0, int64_t graph0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const bool real = i < n;
    pos_out[3 * i] = real ? pos[3 * i] : pad0 + spacing * (T)(i - n);
    pos_out[3 * i + 1] = real ? pos[3 * i + 1] : T(0);
    pos_out[3 * i + 2] = real ? pos[3 * i + 2] : T(0);
    z_out[i] = real ? (int32_t)z[i] : 0;
    int64_t b = G - 1;
    if (real) {
      if (batch) {
        b = batch[i] - graph0;
      } else {   // no graph ids handed over: the graph whose [ptr[k], ptr[k + 1]) holds the atom (empty graphs are skipped: the last k with ptr[k] <= atom)
        int64_t lo = 0, hi = g;   // invariant: ptr[lo] <= atom < ptr[hi]
        const int64_t atom = i + atom0;
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if (ptr[mid] <= atom) lo = mid;
          else hi = mid;
        }
        b = lo;
      }
    }
    batch_out[i] = b;
  }
  if (i <= G) ptr_out[i] = i <= g ? ptr[i] - atom0 : (i < G ? n : N);   // graphs g .. G - 2 are empty, graph G - 1 holds the padding atoms
}
}  // namespace xeq

static int load_padded_batch(int dtype, const void* pos, const void* z, int z_is_int64, const int64_t* ptr, const int64_t* batch, int64_t n,
                             int64_t g, int64_t n_cap, int64_t g_cap, double pad0, double spacing, void* pos_out,
                             int32_t* z_out, int64_t* ptr_out, int64_t* batch_out, void* stream, int64_t atom0 = 0, int64_t graph0 = 0) {
  XEQ_CHECK_ARG(n >= 0 && g >= 0 && n <= n_cap && g < g_cap, "xeq_load_padded_batch: %lld atoms / %lld graphs into a capacity of %lld / %lld (one graph is the padding's)",
                (long long)n, (long long)g, (long long)n_cap, (long long)(g_cap - 1));
  const int64_t threads = (n_cap > g_cap + 1 ? n_cap : g_cap + 1);
  XEQ_DISPATCH_FLOAT(dtype, {
    if (z_is_int64)
      hipLaunchKernelGGL((xeq::k_load_padded_batch<T, int64_t>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)pos, (const int64_t*)z, ptr, batch, n, g, n_cap, g_cap, (T)pad0, (T)spacing, (T*)pos_out, z_out, ptr_out, batch_out,
                         atom0, graph0);
    else
      hipLaunchKernelGGL((xeq::k_load_padded_batch<T, int32_t>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)pos, (const int32_t*)z, ptr, batch, n, g, n_cap, g_cap, (T)pad0, (T)spacing, (T*)pos_out, z_out, ptr_out, batch_out,
                         atom0, graph0);
  });
  XEQ_CHECK_LAUNCH("xeq_load_padded_batch");
  return XEQ_OK;
}

extern "C" int xeq_load_padded_batch(int dtype, const void* pos, const int32_t* z, const int64_t* ptr, const int64_t* batch, int64_t n,
                                     int64_t g, int64_t n_cap, int64_t g_cap, double pad0, double spacing, void* pos_out,
                                     int32_t* z_out, int64_t* ptr_out, int64_t* batch_out, void* stream) {
  return load_padded_batch(dtype, pos, z, 0, ptr, batch, n, g, n_cap, g_cap, pad0, spacing, pos_out, z_out, ptr_out, batch_out, stream);
}
/* the same with int64 atomic numbers (what a torch.long tensor holds): no conversion launch in front of every step */
extern "C" int xeq_load_padded_batch_z64(int dtype, const void* pos, const int64_t* z, const int64_t* ptr, const int64_t* batch, int64_t n,
                                         int64_t g, int64_t n_cap, int64_t g_cap, double pad0, double spacing, void* pos_out,
                                         int32_t* z_out, int64_t* ptr_out, int64_t* batch_out, void* stream) {
  return load_padded_batch(dtype, pos, z, 1, ptr, batch, n, g, n_cap, g_cap, pad0, spacing, pos_out, z_out, ptr_out, batch_out, stream);
}

// ---- the first message block's front half gathered from the element table (xeq_first_block_front, include/xeq.h) ----
namespace xeq {
template <typename Z>
__global__ void k_first_block_front(const Z* __restrict__ z, int64_t n, int64_t n_rows, const float* __restrict__ rows_s, const float* __restrict__ rows_h,
                                    const float* __restrict__ rows_x0, int F, int H, int64_t D, float* __restrict__ s_out,
                                    float* __restrict__ h_out, float* __restrict__ xhat_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 of the three outputs laid end to end
  const int64_t f4 = F >> 2, h4 = H >> 2, d4 = D >> 2;
  const int64_t n_s = n * f4, n_h = n * h4, n_x = n * d4;
  // (an atomic number outside the table reads row 0 -- the table's all-zero padding row -- instead of faulting; the reference's lookup raises)
  auto row_of = [&](int64_t i) { const int64_t r = (int64_t)z[i]; return r >= 0 && r < n_rows ? r : 0; };
  if (t < n_s) {
    const int64_t i = t / f4, c = t - i * f4;
    reinterpret_cast<float4*>(s_out)[t] = reinterpret_cast<const float4*>(rows_s)[row_of(i) * f4 + c];
  } else if (t < n_s + n_h) {
    const int64_t u = t - n_s, i = u / h4, c = u - i * h4;
    reinterpret_cast<float4*>(h_out)[u] = reinterpret_cast<const float4*>(rows_h)[row_of(i) * h4 + c];
  } else if (t < n_s + n_h + n_x) {
    const int64_t u = t - n_s - n_h;   // BT layout: the 0e block [n, F] first, every l > 0 block behind it is the norm of zero: zero
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u < n_s) {
      const int64_t i = u / f4, c = u - i * f4;
      v = reinterpret_cast<const float4*>(rows_x0)[row_of(i) * f4 + c];
    }
    reinterpret_cast<float4*>(xhat_out)[u] = v;
  }
}
}  // namespace xeq

extern "C" int xeq_first_block_front(const void* z, int z_is_int64, int64_t n, int64_t n_rows, const void* rows_s, const void* rows_h, const void* rows_x0,
                                     int node_dim, int hidden_dim, int64_t irreps_dim, void* s_out, void* h_out, void* xhat_out, void* stream) {
  XEQ_CHECK_ARG(n >= 0 && node_dim > 0 && node_dim % 4 == 0 && hidden_dim % 4 == 0 && irreps_dim % 4 == 0 && irreps_dim >= node_dim,
                "xeq_first_block_front: widths must be multiples of four floats");
  XEQ_CHECK_ARG(z && rows_s && rows_h && rows_x0 && s_out && h_out && xhat_out && n_rows >= 1, "xeq_first_block_front: NULL argument / empty table");
  if (n == 0) return XEQ_OK;
  const int64_t total = n * ((node_dim + hidden_dim + irreps_dim) / 4);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (z_is_int64)
    hipLaunchKernelGGL((xeq::k_first_block_front<int64_t>), grid, dim3(256), 0, (hipStream_t)stream, (const int64_t*)z, n, n_rows, (const float*)rows_s,
                       (const float*)rows_h, (const float*)rows_x0, node_dim, hidden_dim, irreps_dim, (float*)s_out, (float*)h_out, (float*)xhat_out);
  else
    hipLaunchKernelGGL((xeq::k_first_block_front<int32_t>), grid, dim3(256), 0, (hipStream_t)stream, (const int32_t*)z, n, n_rows, (const float*)rows_s,
                       (const float*)rows_h, (const float*)rows_x0, node_dim, hidden_dim, irreps_dim, (float*)s_out, (float*)h_out, (float*)xhat_out);
  XEQ_CHECK_LAUNCH("xeq_first_block_front");
  return XEQ_OK;
}
/* a contiguous range of molecules of a larger batch (a shard / lane): pos, z, batch point at the range's first atom, ptr at its first
 * graph; the range's own numbering starts at zero (ptr values - atom_offset, batch values - graph_offset) */
extern "C" int xeq_load_padded_shard(int dtype, const void* pos, const void* z, int z_is_int64, const int64_t* ptr, const int64_t* batch,
                                     int64_t n, int64_t g, int64_t atom_offset, int64_t graph_offset, int64_t n_cap, int64_t g_cap, double pad0,
                                     double spacing, void* pos_out, int32_t* z_out, int64_t* ptr_out, int64_t* batch_out, void* stream) {
  return load_padded_batch(dtype, pos, z, z_is_int64, ptr, batch, n, g, n_cap, g_cap, pad0, spacing, pos_out, z_out, ptr_out, batch_out, stream,
                           atom_offset, graph_offset);
}
