// Neighbour-list and CSR kernels.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
#include <cmath>

#include <hipcub/hipcub.hpp>

#include "xeq_common.h"

namespace xeq {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<int64_t> g_launches{0};
constexpr int64_t LAUNCH_RING = 8192;            // names of the last launches (pointers to string literals), for xeq_launch_names
static const char* g_launch_ring[LAUNCH_RING];
void note_launch(const char* name) { g_launch_ring[g_launches.fetch_add(1, std::memory_order_relaxed) & (LAUNCH_RING - 1)] = name; }

// ---------------------------------------------------------------- CSR helpers
__global__ void k_csr_rowptr(const int64_t* __restrict__ keys, int64_t n_keys, int64_t n_rows,
                             int32_t* __restrict__ rowptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  int64_t lo = 0, hi = n_keys;  // first p with keys[p] >= i
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  rowptr[i] = (int32_t)lo;
}

// keys64 -> int32 keys + identity values (input of the stable radix sort of xeq_csr_by_key)
__global__ void k_sort_prepare(const int64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ k32, int32_t* __restrict__ iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  k32[i] = (int32_t)keys[i];
  iota[i] = (int32_t)i;
}
// capacity form: only the first *n_valid keys are edges; the slots behind them sort to the end as row n_rows
__global__ void k_sort_prepare_bounded(const int64_t* __restrict__ keys, int64_t n, const int32_t* __restrict__ n_valid, int32_t n_rows,
                                       int32_t* __restrict__ k32, int32_t* __restrict__ iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  k32[i] = i < (int64_t)n_valid[0] ? (int32_t)keys[i] : n_rows;
  iota[i] = (int32_t)i;
}
__global__ void k_csr_rowptr32(const int32_t* __restrict__ keys, int64_t n_keys, int64_t n_rows, int32_t* __restrict__ rowptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  int64_t lo = 0, hi = n_keys;  // first p with keys[p] >= i
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  rowptr[i] = (int32_t)lo;
}

// single-workgroup scan with carry; n is O(nodes) and this runs once per graph build, in front of the host's
// read-back of the edge count.  Each thread takes 8 consecutive counts, so a round covers 8192 entries.
__global__ void __launch_bounds__(1024) k_exclusive_scan_i32(const int32_t* __restrict__ in, int64_t n,
                                                             int32_t* __restrict__ out) {
  constexpr int IT = 8;
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024 * IT) {
    const int64_t i0 = base + (int64_t)t * IT;
    int32_t v[IT];
    int32_t tot = 0;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      v[k] = i0 + k < n ? in[i0 + k] : 0;
      tot += v[k];
    }
    int32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int32_t u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    const int32_t carry = carry_s;
    int32_t run = carry + woff + incl - tot;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      if (i0 + k < n) out[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (t == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (t == 0) out[n] = carry_s;
}

// n_perm of a SYMMETRIC center-sorted edge list with ascending, unique neighbours per center (what the open-boundary
// builders emit): the neighbour-sorted order of edge (i -> j) is the position of its reverse edge (j -> i) in the
// center-sorted list, found by binary search in segment j.  rev[e] = -1 when the reverse edge does not exist.
__global__ void k_reverse_edge_map(const int64_t* __restrict__ center, const int64_t* __restrict__ nbr,
                                   const int32_t* __restrict__ c_rowptr, int64_t E, int64_t N, int32_t* __restrict__ rev) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E || e >= c_rowptr[N]) return;   // E may be a capacity: the list itself ends at c_rowptr[N] (device-side count)
  const int64_t i = center[e], j = nbr[e];
  int32_t lo = c_rowptr[j], hi = c_rowptr[j + 1];
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (nbr[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  rev[e] = (lo < c_rowptr[j + 1] && nbr[lo] == i) ? lo : -1;
}

// The same map for a PERIODIC list of this library's builders: center-sorted, a center's edges ascending in (neighbor, image index), and
// (i <- j, offset o) present iff (j <- i, -o) is -- up to a rounding at the cutoff (the distance of the two is formed from differently
// rounded sums, data/radius_graph.py:117-121, so one of a pair may fall on the other side of `<` when it sits within an ulp of the
// cutoff; there the envelope is ~1e-14).  Binary search for the first slot of row j whose neighbor is i, then the run of that neighbor
// for the slot with the negated offset (offsets are whole numbers held in floating point: exact).  rev[e] = -1: no such edge.
template <typename T>
__global__ void k_reverse_edge_map_pbc(const int64_t* __restrict__ center, const int64_t* __restrict__ nbr, const T* __restrict__ off,
                                       const int32_t* __restrict__ c_rowptr, int64_t E, int64_t N, int32_t* __restrict__ rev) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t count = c_rowptr[N] < E ? (int64_t)c_rowptr[N] : E;   // E may be a capacity (a list cut at it keeps its true count in c_rowptr[N])
  if (e >= count) return;
  const int64_t i = center[e], j = nbr[e];
  const T o0 = -off[3 * e], o1 = -off[3 * e + 1], o2 = -off[3 * e + 2];
  const int32_t end = c_rowptr[j + 1] < count ? c_rowptr[j + 1] : (int32_t)count;
  int32_t lo = c_rowptr[j], hi = end;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (nbr[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  int32_t r = -1;
  for (int32_t q = lo; q < end && nbr[q] == i; ++q)
    if (off[3 * (int64_t)q] == o0 && off[3 * (int64_t)q + 1] == o1 && off[3 * (int64_t)q + 2] == o2) {
      r = q;
      break;
    }
  rev[e] = r;
}

// ---------------------------------------------------------- non-PBC radius graph
// One thread per center; all lanes of a wave walk (mostly) the same molecule, so the
// position loads broadcast.  d^2 is evaluated without fma contraction so that the
// edge decisions are those of the oracle's  (dx*dx + dy*dy) + dz*dz  in the same dtype.
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b);
template <> __device__ __forceinline__ float mul_rn<float>(float a, float b) { return __fmul_rn(a, b); }
template <> __device__ __forceinline__ double mul_rn<double>(double a, double b) { return __dmul_rn(a, b); }
template <typename T> __device__ __forceinline__ T add_rn(T a, T b);
template <> __device__ __forceinline__ float add_rn<float>(float a, float b) { return __fadd_rn(a, b); }
template <> __device__ __forceinline__ double add_rn<double>(double a, double b) { return __dadd_rn(a, b); }
template <typename T> __device__ __forceinline__ T sub_rn(T a, T b);
template <> __device__ __forceinline__ float sub_rn<float>(float a, float b) { return __fsub_rn(a, b); }
template <> __device__ __forceinline__ double sub_rn<double>(double a, double b) { return __dsub_rn(a, b); }

__device__ __forceinline__ int64_t graph_of(const int64_t* __restrict__ ptr, int64_t n_graphs, int64_t i) {
  int64_t lo = 0, hi = n_graphs;  // last g with ptr[g] <= i
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid;
    else hi = mid;
  }
  return lo;
}

// One wave per center, lane = candidate neighbour (64 at a time in ascending index: a ballot + prefix popcount keeps the
// (center, neighbor) order).  A thread per center walking its molecule in a dependent loop took 13 + 7 us (count + fill) for
// the 335 k distance checks of QM9-1024 on a third of the CUs.
template <typename T, bool FILL>
__global__ void k_radius_graph(const T* __restrict__ pos, const int64_t* __restrict__ ptr, int64_t n_graphs,
                               int64_t n_nodes, T r2, int32_t* __restrict__ deg,
                               const int32_t* __restrict__ rowptr, int64_t n_edges,
                               int64_t* __restrict__ edge_index) {
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  const int64_t g = graph_of(ptr, n_graphs, i);
  const int64_t a = ptr[g], b = ptr[g + 1];
  const T xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  int32_t cnt = 0;
  int64_t w = FILL ? (int64_t)rowptr[i] : 0;
  for (int64_t j0 = a; j0 < b; j0 += 64) {
    const int64_t j = j0 + lane;
    bool hit = false;
    if (j < b) {
      T dx = sub_rn<T>(xi, pos[3 * j]), dy = sub_rn<T>(yi, pos[3 * j + 1]), dz = sub_rn<T>(zi, pos[3 * j + 2]);
      T d2 = add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz));
      hit = d2 < r2 && j != i;
    }
    const unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      const int64_t p = w + __popcll(m & ((1ull << lane) - 1ull));
      if (p < n_edges) {              // n_edges: the row length of edge_index -- the edge count, or a capacity (never written past)
        edge_index[p] = i;            // center
        edge_index[n_edges + p] = j;  // neighbor
      }
    }
    const int pc = __popcll(m);
    w += pc;
    cnt += pc;
  }
  if (!FILL && lane == 0) deg[i] = cnt;
}

// Wrap positions into the unit cell (data/radius_graph.py:6-32): fractional = pos cell^-1, shift = floor(fractional) on the
// periodic axes, pos_wrap = (fractional - shift) cell.  One thread per atom; the per-graph inverse comes from the host (the cell
// is a handful of numbers: its inverse, the image counts and the image table are formed there in one round trip instead of
// ~40 tiny device launches).
template <typename T>
__global__ void k_pbc_wrap(const T* __restrict__ pos, const int64_t* __restrict__ ptr, int64_t n_graphs, int64_t n_nodes,
                           const T* __restrict__ cell, const T* __restrict__ cell_inv, int px, int py, int pz,
                           T* __restrict__ pos_wrap, T* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int64_t g = graph_of(ptr, n_graphs, i);
  const T* M = cell + 9 * g;
  const T* Mi = cell_inv + 9 * g;
  const T p0 = pos[3 * i], p1 = pos[3 * i + 1], p2 = pos[3 * i + 2];
  const int per[3] = {px, py, pz};
  T f[3], s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f[k] = add_rn<T>(add_rn<T>(mul_rn<T>(p0, Mi[k]), mul_rn<T>(p1, Mi[3 + k])), mul_rn<T>(p2, Mi[6 + k]));
    s[k] = per[k] ? floor(f[k]) : T(0);
    f[k] = sub_rn<T>(f[k], s[k]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pos_wrap[3 * i + k] = add_rn<T>(add_rn<T>(mul_rn<T>(f[0], M[k]), mul_rn<T>(f[1], M[3 + k])), mul_rn<T>(f[2], M[6 + k]));
    shift[3 * i + k] = s[k];
  }
}

// -------------------------------------------------------------- PBC radius graph
// One wave per center.  Candidate keys k = (j - a) * n_cells + c are swept in
// ascending order 64 at a time; a ballot + prefix popcount keeps the reference's
// (neighbor * n_cells + cell) ordering without any sort.
template <typename T, bool FILL>
__global__ void k_radius_graph_pbc(const T* __restrict__ pw, const int64_t* __restrict__ ptr, int64_t n_graphs,
                                   int64_t n_nodes, const T* __restrict__ img, const T* __restrict__ cells,
                                   const T* __restrict__ shift, int64_t n_cells, T rc, int32_t* __restrict__ deg,
                                   const int32_t* __restrict__ rowptr, int64_t n_edges,
                                   int64_t* __restrict__ edge_index, T* __restrict__ cell_offsets) {
  const int lane = threadIdx.x & 63;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  int64_t g = graph_of(ptr, n_graphs, i);
  int64_t a = ptr[g], b = ptr[g + 1];
  const T* gimg = img + g * n_cells * 3;
  T xi = pw[3 * i], yi = pw[3 * i + 1], zi = pw[3 * i + 2];
  int64_t n_keys = (b - a) * n_cells;
  int64_t w = FILL ? (int64_t)rowptr[i] : 0;
  int32_t cnt = 0;
  for (int64_t k0 = 0; k0 < n_keys; k0 += 64) {
    int64_t k = k0 + lane;
    bool hit = false;
    int64_t j = 0;
    int64_t c = 0;
    if (k < n_keys) {
      j = a + k / n_cells;
      c = k % n_cells;
      // B = pos_wrap[j] + img[c]  (radius_graph.py:117), D = |A - B| (cdist)
      T bx = add_rn<T>(pw[3 * j], gimg[3 * c]), by = add_rn<T>(pw[3 * j + 1], gimg[3 * c + 1]),
        bz = add_rn<T>(pw[3 * j + 2], gimg[3 * c + 2]);
      T dx = sub_rn<T>(xi, bx), dy = sub_rn<T>(yi, by), dz = sub_rn<T>(zi, bz);
      T D = sqrt_<T>(add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz)));
      hit = (D < rc) && (D > T(0.01));
    }
    unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      int64_t p = w + __popcll(m & ((1ull << lane) - 1ull));
      edge_index[p] = i;
      edge_index[n_edges + p] = j;
      for (int ax = 0; ax < 3; ++ax)
        cell_offsets[3 * p + ax] = cells[3 * c + ax] + (shift[3 * i + ax] - shift[3 * j + ax]);
    }
    int pc = __popcll(m);
    w += pc;
    cnt += pc;
  }
  if (!FILL && lane == 0) deg[i] = cnt;
}

// Image-pruned form of the same search (default).  One wave per center, lane = candidate neighbor j (64 at a time, in
// ascending order).  Instead of testing all n_cells images of every j, the lane derives from the fractional
// separation f = (A - pos_wrap[j]) . recip which image offsets n can possibly be within the cutoff: along axis a the
// distance is at least |f_a - n_a| / |recip_a| (distance between lattice planes), so only ceil(f_a - t_a) <= n_a <=
// floor(f_a + t_a), t_a = rc |recip_a| + margin, survive -- typically one image instead of 27.  The survivors are
// evaluated with EXACTLY the arithmetic of k_radius_graph_pbc (so the edge set is bit-identical), in ascending cell
// index; a wave prefix over the lanes' hit counts keeps the reference's (neighbor * n_cells + cell) order.
struct PbcPrune {
  int rep[3];   // images per axis: cell index c = ((n0 + rep0) (2 rep1 + 1) + (n1 + rep1)) (2 rep2 + 1) + (n2 + rep2)
};
template <typename T>
__device__ __forceinline__ T floor_(T x);
template <> __device__ __forceinline__ float floor_<float>(float x) { return floorf(x); }
template <> __device__ __forceinline__ double floor_<double>(double x) { return floor(x); }
template <typename T>
__device__ __forceinline__ T ceil_(T x);
template <> __device__ __forceinline__ float ceil_<float>(float x) { return ceilf(x); }
template <> __device__ __forceinline__ double ceil_<double>(double x) { return ceil(x); }

template <typename T, bool FILL>
__global__ void k_radius_graph_pbc_img(const T* __restrict__ pw, const int64_t* __restrict__ ptr, int64_t n_graphs,
                                       int64_t n_nodes, const T* __restrict__ img, const T* __restrict__ cells,
                                       const T* __restrict__ shift, int64_t n_cells, T rc, const T* __restrict__ recip,
                                       const T* __restrict__ thr, PbcPrune pr, int32_t* __restrict__ deg,
                                       const int32_t* __restrict__ rowptr, int64_t n_edges,
                                       int64_t* __restrict__ edge_index, T* __restrict__ cell_offsets) {
  const int lane = threadIdx.x & 63;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  int64_t g = graph_of(ptr, n_graphs, i);
  int64_t a = ptr[g], b = ptr[g + 1];
  const T* gimg = img + g * n_cells * 3;
  const T* rg = recip + g * 9;
  const T t0 = thr[3 * g], t1 = thr[3 * g + 1], t2 = thr[3 * g + 2];
  const T xi = pw[3 * i], yi = pw[3 * i + 1], zi = pw[3 * i + 2];
  const int w1 = 2 * pr.rep[1] + 1, w2 = 2 * pr.rep[2] + 1;
  int64_t w = FILL ? (int64_t)rowptr[i] : 0;
  int32_t cnt = 0;
  for (int64_t j0 = a; j0 < b; j0 += 64) {
    const int64_t j = j0 + lane;
    const bool have = j < b;
    int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};   // empty ranges for lanes without a candidate
    T xj = T(0), yj = T(0), zj = T(0);
    if (have) {
      xj = pw[3 * j];
      yj = pw[3 * j + 1];
      zj = pw[3 * j + 2];
      const T vx = xi - xj, vy = yi - yj, vz = zi - zj;
      const T f[3] = {vx * rg[0] + vy * rg[1] + vz * rg[2], vx * rg[3] + vy * rg[4] + vz * rg[5],
                      vx * rg[6] + vy * rg[7] + vz * rg[8]};
      const T t[3] = {t0, t1, t2};
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        if (pr.rep[ax] == 0) {
          lo[ax] = hi[ax] = 0;
        } else {
          const T l = ceil_<T>(f[ax] - t[ax]), h = floor_<T>(f[ax] + t[ax]);
          lo[ax] = l < T(-pr.rep[ax]) ? -pr.rep[ax] : (int)l;
          hi[ax] = h > T(pr.rep[ax]) ? pr.rep[ax] : (int)h;
        }
      }
    }
    // ---- pass 1: this lane's hits (cells in ascending index)
    int nh = 0;
    for (int n0 = lo[0]; n0 <= hi[0]; ++n0)
      for (int n1 = lo[1]; n1 <= hi[1]; ++n1)
        for (int n2 = lo[2]; n2 <= hi[2]; ++n2) {
          const int64_t c = ((int64_t)(n0 + pr.rep[0]) * w1 + (n1 + pr.rep[1])) * w2 + (n2 + pr.rep[2]);
          T bx = add_rn<T>(xj, gimg[3 * c]), by = add_rn<T>(yj, gimg[3 * c + 1]), bz = add_rn<T>(zj, gimg[3 * c + 2]);
          T dx = sub_rn<T>(xi, bx), dy = sub_rn<T>(yi, by), dz = sub_rn<T>(zi, bz);
          T D = sqrt_<T>(add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz)));
          if ((D < rc) && (D > T(0.01))) ++nh;
        }
    // exclusive prefix of nh over the lanes, bit by bit (nh is small)
    int base = 0, total = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int bit = 0; bit < 31; ++bit) {
      const unsigned long long m = __ballot((nh >> bit) & 1);
      base += __popcll(m & lt) << bit;
      total += __popcll(m) << bit;
      if (__ballot(nh >> (bit + 1)) == 0ull) break;
    }
    if (FILL && nh > 0) {   // ---- pass 2: the same walk, writing
      int64_t p = w + base;
      for (int n0 = lo[0]; n0 <= hi[0]; ++n0)
        for (int n1 = lo[1]; n1 <= hi[1]; ++n1)
          for (int n2 = lo[2]; n2 <= hi[2]; ++n2) {
            const int64_t c = ((int64_t)(n0 + pr.rep[0]) * w1 + (n1 + pr.rep[1])) * w2 + (n2 + pr.rep[2]);
            T bx = add_rn<T>(xj, gimg[3 * c]), by = add_rn<T>(yj, gimg[3 * c + 1]), bz = add_rn<T>(zj, gimg[3 * c + 2]);
            T dx = sub_rn<T>(xi, bx), dy = sub_rn<T>(yi, by), dz = sub_rn<T>(zi, bz);
            T D = sqrt_<T>(add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz)));
            if ((D < rc) && (D > T(0.01))) {
              if (p < n_edges) {   // (capacity form: a list that outgrew its buffers is cut, the row pointer still tells its size)
                edge_index[p] = i;
                edge_index[n_edges + p] = j;
                for (int ax = 0; ax < 3; ++ax)
                  cell_offsets[3 * p + ax] = cells[3 * c + ax] + (shift[3 * i + ax] - shift[3 * j + ax]);
              }
              ++p;
            }
          }
    }
    w += total;
    cnt += total;
  }
  if (!FILL && lane == 0) deg[i] = cnt;
}

// ------------------------------------------------------------------------------ PBC radius graph, cell list
// For graphs of many atoms the O(n_g^2) pair sweep above is replaced by a bin grid in fractional coordinates:
//   bins per periodic axis nb_a = floor(1 / thr_a) (bin width >= the cutoff measured across lattice planes), so every
//   neighbor of a center lies in the 3 x 3 x 3 block of bins around it (periodic wrap; axes with nb <= 2 visit each
//   bin once; open axes have one bin);
//   atoms are sorted by (graph, bin) with xeq_csr_by_key (stable: ascending atom index inside a bin);
//   one wave per center walks the <= 27 bins, lane = candidate atom, and evaluates the surviving images with exactly
//   the arithmetic of k_radius_graph_pbc;
//   hits are written unordered as keys (j - a) n_cells + c into the center's segment and k_pbc_sort_segment ranks them
//   (keys are unique per center), which restores the reference's (neighbor * n_cells + cell) order bit for bit.
template <typename T>
__device__ __forceinline__ void pbc_bin_of(const T* __restrict__ p, const T* __restrict__ rg, const int32_t* __restrict__ nb,
                                           int (&b)[3]) {
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const T f = p[0] * rg[3 * ax] + p[1] * rg[3 * ax + 1] + p[2] * rg[3 * ax + 2];
    int v = (int)floor_<T>(f * T(nb[ax]));
    v = v < 0 ? 0 : (v >= nb[ax] ? nb[ax] - 1 : v);   // wrapped positions sit in [0, 1) up to rounding
    b[ax] = nb[ax] > 1 ? v : 0;
  }
}

template <typename T>
__global__ void k_pbc_bin_ids(const T* __restrict__ pw, const int64_t* __restrict__ ptr, int64_t n_graphs, int64_t n_nodes,
                              const T* __restrict__ recip, const int32_t* __restrict__ nb, const int32_t* __restrict__ bin_base,
                              int64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int64_t g = graph_of(ptr, n_graphs, i);
  int b[3];
  pbc_bin_of<T>(pw + 3 * i, recip + 9 * g, nb + 3 * g, b);
  keys[i] = (int64_t)bin_base[g] + ((int64_t)b[0] * nb[3 * g + 1] + b[1]) * nb[3 * g + 2] + b[2];
}

template <typename T, bool FILL>
__global__ void k_radius_graph_pbc_cl(const T* __restrict__ pw, const int64_t* __restrict__ ptr, int64_t n_graphs,
                                      int64_t n_nodes, const T* __restrict__ img, int64_t n_cells, T rc,
                                      const T* __restrict__ recip, const T* __restrict__ thr, PbcPrune pr,
                                      const int32_t* __restrict__ nb, const int32_t* __restrict__ bin_base,
                                      const int32_t* __restrict__ bin_start, const int32_t* __restrict__ bin_atom,
                                      int32_t* __restrict__ deg, const int32_t* __restrict__ rowptr,
                                      int64_t* __restrict__ tmp_keys) {
  const int lane = threadIdx.x & 63;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  const int64_t g = graph_of(ptr, n_graphs, i);
  const int64_t a = ptr[g];
  const T* gimg = img + g * n_cells * 3;
  const T* rg = recip + g * 9;
  const int32_t* nbg = nb + 3 * g;
  const T t[3] = {thr[3 * g], thr[3 * g + 1], thr[3 * g + 2]};
  const T xi = pw[3 * i], yi = pw[3 * i + 1], zi = pw[3 * i + 2];
  const int w1 = 2 * pr.rep[1] + 1, w2 = 2 * pr.rep[2] + 1;
  int bi[3];
  pbc_bin_of<T>(pw + 3 * i, rg, nbg, bi);
  int cntax[3], first[3];   // bins to visit per axis: all of them when nb <= 2, else b-1, b, b+1 (wrapped)
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    cntax[ax] = nbg[ax] <= 2 ? nbg[ax] : 3;
    first[ax] = nbg[ax] <= 2 ? 0 : bi[ax] - 1;
  }
  int64_t w = FILL ? (int64_t)rowptr[i] : 0;
  int32_t cnt = 0;
  for (int q0 = 0; q0 < cntax[0]; ++q0)
    for (int q1 = 0; q1 < cntax[1]; ++q1)
      for (int q2 = 0; q2 < cntax[2]; ++q2) {
        int b0 = first[0] + q0, b1 = first[1] + q1, b2 = first[2] + q2;
        b0 = b0 < 0 ? b0 + nbg[0] : (b0 >= nbg[0] ? b0 - nbg[0] : b0);
        b1 = b1 < 0 ? b1 + nbg[1] : (b1 >= nbg[1] ? b1 - nbg[1] : b1);
        b2 = b2 < 0 ? b2 + nbg[2] : (b2 >= nbg[2] ? b2 - nbg[2] : b2);
        const int64_t bin = (int64_t)bin_base[g] + ((int64_t)b0 * nbg[1] + b1) * nbg[2] + b2;
        const int s0 = bin_start[bin], s1 = bin_start[bin + 1];
        for (int sl0 = s0; sl0 < s1; sl0 += 64) {
          const int sl = sl0 + lane;
          const bool have = sl < s1;
          const int64_t j = have ? (int64_t)bin_atom[sl] : a;
          int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
          T xj = T(0), yj = T(0), zj = T(0);
          if (have) {
            xj = pw[3 * j];
            yj = pw[3 * j + 1];
            zj = pw[3 * j + 2];
            const T vx = xi - xj, vy = yi - yj, vz = zi - zj;
            const T f[3] = {vx * rg[0] + vy * rg[1] + vz * rg[2], vx * rg[3] + vy * rg[4] + vz * rg[5],
                            vx * rg[6] + vy * rg[7] + vz * rg[8]};
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              if (pr.rep[ax] == 0) {
                lo[ax] = hi[ax] = 0;
              } else {
                const T l = ceil_<T>(f[ax] - t[ax]), h = floor_<T>(f[ax] + t[ax]);
                lo[ax] = l < T(-pr.rep[ax]) ? -pr.rep[ax] : (int)l;
                hi[ax] = h > T(pr.rep[ax]) ? pr.rep[ax] : (int)h;
              }
            }
          }
          int nh = 0;
          for (int n0 = lo[0]; n0 <= hi[0]; ++n0)
            for (int n1 = lo[1]; n1 <= hi[1]; ++n1)
              for (int n2 = lo[2]; n2 <= hi[2]; ++n2) {
                const int64_t c = ((int64_t)(n0 + pr.rep[0]) * w1 + (n1 + pr.rep[1])) * w2 + (n2 + pr.rep[2]);
                T bx = add_rn<T>(xj, gimg[3 * c]), by = add_rn<T>(yj, gimg[3 * c + 1]), bz = add_rn<T>(zj, gimg[3 * c + 2]);
                T dx = sub_rn<T>(xi, bx), dy = sub_rn<T>(yi, by), dz = sub_rn<T>(zi, bz);
                T D = sqrt_<T>(add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz)));
                if ((D < rc) && (D > T(0.01))) ++nh;
              }
          int base = 0, total = 0;
          const unsigned long long lt = (1ull << lane) - 1ull;
          for (int bit = 0; bit < 31; ++bit) {
            const unsigned long long m = __ballot((nh >> bit) & 1);
            base += __popcll(m & lt) << bit;
            total += __popcll(m) << bit;
            if (__ballot(nh >> (bit + 1)) == 0ull) break;
          }
          if (FILL && nh > 0) {
            int64_t p = w + base;
            for (int n0 = lo[0]; n0 <= hi[0]; ++n0)
              for (int n1 = lo[1]; n1 <= hi[1]; ++n1)
                for (int n2 = lo[2]; n2 <= hi[2]; ++n2) {
                  const int64_t c = ((int64_t)(n0 + pr.rep[0]) * w1 + (n1 + pr.rep[1])) * w2 + (n2 + pr.rep[2]);
                  T bx = add_rn<T>(xj, gimg[3 * c]), by = add_rn<T>(yj, gimg[3 * c + 1]), bz = add_rn<T>(zj, gimg[3 * c + 2]);
                  T dx = sub_rn<T>(xi, bx), dy = sub_rn<T>(yi, by), dz = sub_rn<T>(zi, bz);
                  T D = sqrt_<T>(add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz)));
                  if ((D < rc) && (D > T(0.01))) tmp_keys[p++] = (j - a) * n_cells + c;
                }
          }
          w += total;
          cnt += total;
        }
      }
  if (!FILL && lane == 0) deg[i] = cnt;
}

// rank the unordered keys of every center and emit edge_index / cell_offsets in ascending key order
template <typename T>
__global__ void k_pbc_sort_segment(const int64_t* __restrict__ tmp_keys, const int32_t* __restrict__ rowptr,
                                   const int64_t* __restrict__ ptr, int64_t n_graphs, int64_t n_nodes, int64_t n_cells,
                                   const T* __restrict__ cells, const T* __restrict__ shift, int64_t n_edges,
                                   int64_t* __restrict__ edge_index, T* __restrict__ cell_offsets) {
  const int lane = threadIdx.x & 63;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  const int64_t a = ptr[graph_of(ptr, n_graphs, i)];
  const int s0 = rowptr[i], s1 = rowptr[i + 1];
  for (int q = s0 + lane; q < s1; q += 64) {
    const int64_t key = tmp_keys[q];
    int rank = 0;
    for (int r = s0; r < s1; ++r) rank += tmp_keys[r] < key ? 1 : 0;
    const int64_t p = (int64_t)s0 + rank, jl = key / n_cells, c = key - jl * n_cells, j = a + jl;
    edge_index[p] = i;
    edge_index[n_edges + p] = j;
    for (int ax = 0; ax < 3; ++ax) cell_offsets[3 * p + ax] = cells[3 * c + ax] + (shift[3 * i + ax] - shift[3 * j + ax]);
  }
}

// ------------------------------------------------------------------------------ open-boundary radius graph, cell list
// Same idea without images: per graph an axis-aligned grid over its bounding box (lo[G,3], inverse bin width
// inv_w[G,3], nbins[G,3]; bin width >= the cutoff, so all neighbors of a center sit in the 3x3x3 block around its bin).
// The per-pair test is the arithmetic of k_radius_graph (d^2 < r^2, rounding-explicit), hits are ranked per center, so
// edge_index is the same canonical (center, neighbor) order, bit for bit.
template <typename T>
__device__ __forceinline__ void box_bin_of(const T* __restrict__ p, const T* __restrict__ lo, const T* __restrict__ inv_w,
                                           const int32_t* __restrict__ nb, int (&b)[3]) {
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    int v = (int)floor_<T>((p[ax] - lo[ax]) * inv_w[ax]);
    b[ax] = v < 0 ? 0 : (v >= nb[ax] ? nb[ax] - 1 : v);
  }
}

template <typename T>
__global__ void k_box_bin_ids(const T* __restrict__ pos, const int64_t* __restrict__ ptr, int64_t n_graphs, int64_t n_nodes,
                              const T* __restrict__ lo, const T* __restrict__ inv_w, const int32_t* __restrict__ nb,
                              const int32_t* __restrict__ bin_base, int64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const int64_t g = graph_of(ptr, n_graphs, i);
  int b[3];
  box_bin_of<T>(pos + 3 * i, lo + 3 * g, inv_w + 3 * g, nb + 3 * g, b);
  keys[i] = (int64_t)bin_base[g] + ((int64_t)b[0] * nb[3 * g + 1] + b[1]) * nb[3 * g + 2] + b[2];
}

template <typename T, bool FILL>
__global__ void k_radius_graph_cl(const T* __restrict__ pos, const int64_t* __restrict__ ptr, int64_t n_graphs,
                                  int64_t n_nodes, T r2, const T* __restrict__ lo, const T* __restrict__ inv_w,
                                  const int32_t* __restrict__ nb, const int32_t* __restrict__ bin_base,
                                  const int32_t* __restrict__ bin_start, const int32_t* __restrict__ bin_atom,
                                  int32_t* __restrict__ deg, const int32_t* __restrict__ rowptr,
                                  int64_t* __restrict__ tmp_keys) {
  const int lane = threadIdx.x & 63;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  const int64_t g = graph_of(ptr, n_graphs, i);
  const int32_t* nbg = nb + 3 * g;
  const T xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  int bi[3];
  box_bin_of<T>(pos + 3 * i, lo + 3 * g, inv_w + 3 * g, nbg, bi);
  int64_t w = FILL ? (int64_t)rowptr[i] : 0;
  int32_t cnt = 0;
  for (int b0 = max(bi[0] - 1, 0); b0 <= min(bi[0] + 1, nbg[0] - 1); ++b0)
    for (int b1 = max(bi[1] - 1, 0); b1 <= min(bi[1] + 1, nbg[1] - 1); ++b1)
      for (int b2 = max(bi[2] - 1, 0); b2 <= min(bi[2] + 1, nbg[2] - 1); ++b2) {
        const int64_t bin = (int64_t)bin_base[g] + ((int64_t)b0 * nbg[1] + b1) * nbg[2] + b2;
        const int s0 = bin_start[bin], s1 = bin_start[bin + 1];
        for (int sl0 = s0; sl0 < s1; sl0 += 64) {
          const int sl = sl0 + lane;
          bool hit = false;
          int64_t j = 0;
          if (sl < s1) {
            j = bin_atom[sl];
            T dx = sub_rn<T>(xi, pos[3 * j]), dy = sub_rn<T>(yi, pos[3 * j + 1]), dz = sub_rn<T>(zi, pos[3 * j + 2]);
            T d2 = add_rn<T>(add_rn<T>(mul_rn<T>(dx, dx), mul_rn<T>(dy, dy)), mul_rn<T>(dz, dz));
            hit = d2 < r2 && j != i;
          }
          const unsigned long long m = __ballot(hit);
          if (FILL && hit) tmp_keys[w + __popcll(m & ((1ull << lane) - 1ull))] = j;
          const int pc = __popcll(m);
          w += pc;
          cnt += pc;
        }
      }
  if (!FILL && lane == 0) deg[i] = cnt;
}

// rank the unordered neighbor ids of every center and emit edge_index in (center, neighbor) order
__global__ void k_sort_segment_plain(const int64_t* __restrict__ tmp_keys, const int32_t* __restrict__ rowptr, int64_t n_nodes,
                                     int64_t n_edges, int64_t* __restrict__ edge_index) {
  const int lane = threadIdx.x & 63;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_nodes) return;
  const int s0 = rowptr[i], s1 = rowptr[i + 1];
  for (int q = s0 + lane; q < s1; q += 64) {
    const int64_t key = tmp_keys[q];
    int rank = 0;
    for (int r = s0; r < s1; ++r) rank += tmp_keys[r] < key ? 1 : 0;
    edge_index[(int64_t)s0 + rank] = i;
    edge_index[n_edges + s0 + rank] = key;
  }
}

// degrees -> guarded row pointer in ONE launch (xeq_rowptr_from_degrees): the scan, the total and the capacity guard of
// xeq_rowptr_guard by one workgroup
__global__ void __launch_bounds__(SCAN_WG_THREADS) k_rowptr_from_degrees(const int32_t* __restrict__ deg, int64_t n, int64_t cap,
                                                                        int32_t* __restrict__ rowptr, int32_t* __restrict__ count,
                                                                        int64_t* __restrict__ running_total) {
  extern __shared__ int32_t scan_lds[];
  const int32_t total = wg_scan_lds([&](int64_t i) { return deg[i]; }, n, scan_lds);
  const bool ok = cap < 0 || (int64_t)total <= cap;
#pragma unroll 4
  for (int64_t i = threadIdx.x; i < n; i += SCAN_WG_THREADS) rowptr[i] = ok ? scan_lds[i] : 0;
  if (threadIdx.x == 0) {
    rowptr[n] = ok ? total : 0;
    if (count) count[0] = total;
    if (running_total) running_total[0] += total;   // one thread of one workgroup: a plain read-modify-write
  }
}

__global__ void k_rowptr_guard(const int32_t* __restrict__ raw, int64_t n, int32_t cap, int32_t* __restrict__ rowptr, int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const int32_t total = raw[n];          // (read-only input: no thread writes what another reads)
  if (i == n) count[0] = total;
  rowptr[i] = total > cap ? 0 : raw[i];
}

}  // namespace xeq

using namespace xeq;

// ---- host-side cell tables of the periodic search (xeq_pbc_image_counts / xeq_pbc_tables_host, include/xeq.h) ----
// Every operation is rounded once in T, in the order written (volatile temporaries keep the host compiler from contracting or
// re-associating under -ffp-contract=fast): the Python front and the registered operator both call these, so the search
// kernels see the same bits from either.
namespace xeq {
template <typename T>
struct CellGeom {
  T cross[3][3];   // a_1 x a_2, a_2 x a_0, a_0 x a_1
  T vol;
  T recip[3][3];   // cross[ax] / vol
  T inv_min[3];    // |recip[ax]|
};
template <typename T>
static inline T rnd(T v) {
  volatile T t = v;
  return t;
}
template <typename T>
static void cell_geom(const T* c, CellGeom<T>& g) {   // c: [3][3] row-major, rows = lattice vectors
  const int jj[3] = {1, 2, 0}, kk[3] = {2, 0, 1};
  for (int ax = 0; ax < 3; ++ax) {
    const T* a = c + 3 * jj[ax];
    const T* b = c + 3 * kk[ax];
    g.cross[ax][0] = rnd<T>(rnd<T>(a[1] * b[2]) - rnd<T>(a[2] * b[1]));
    g.cross[ax][1] = rnd<T>(rnd<T>(a[2] * b[0]) - rnd<T>(a[0] * b[2]));
    g.cross[ax][2] = rnd<T>(rnd<T>(a[0] * b[1]) - rnd<T>(a[1] * b[0]));
  }
  T v = rnd<T>(c[0] * g.cross[0][0]);
  v = rnd<T>(v + rnd<T>(c[1] * g.cross[0][1]));
  v = rnd<T>(v + rnd<T>(c[2] * g.cross[0][2]));
  g.vol = v;
  for (int ax = 0; ax < 3; ++ax) {
    T q = T(0);
    for (int j = 0; j < 3; ++j) {
      g.recip[ax][j] = rnd<T>(g.cross[ax][j] / g.vol);
      const T sq = rnd<T>(g.recip[ax][j] * g.recip[ax][j]);
      q = j == 0 ? sq : rnd<T>(q + sq);
    }
    g.inv_min[ax] = rnd<T>(std::sqrt(q));
  }
}
template <typename T>
static void image_counts(const T* cell, int64_t G, const int32_t pbc[3], double cutoff, int32_t reps[3]) {
  for (int ax = 0; ax < 3; ++ax) reps[ax] = 0;
  for (int64_t g = 0; g < G; ++g) {
    CellGeom<T> cg;
    cell_geom<T>(cell + 9 * g, cg);
    for (int ax = 0; ax < 3; ++ax) {
      if (!pbc[ax]) continue;
      const T r = std::ceil(rnd<T>((T)cutoff * cg.inv_min[ax]));
      const int32_t ri = (r > T(0) && r < T(1 << 20)) ? (int32_t)r : (r > T(0) ? (1 << 20) : 0);   // (a degenerate cell: caught by the caller's size check)
      if (ri > reps[ax]) reps[ax] = ri;
    }
  }
}
template <typename T>
static void tables_host(const T* cell, int64_t G, const int32_t reps[3], double cutoff, T* out) {
  const int64_t n0 = 2 * reps[0] + 1, n1 = 2 * reps[1] + 1, n2 = 2 * reps[2] + 1, nc = n0 * n1 * n2;
  T* grid = out;
  T* offs = grid + 3 * nc;
  T* recip = offs + 3 * nc * G;
  T* thr = recip + 9 * G;
  int64_t c = 0;
  for (int64_t i0 = -reps[0]; i0 <= reps[0]; ++i0)
    for (int64_t i1 = -reps[1]; i1 <= reps[1]; ++i1)
      for (int64_t i2 = -reps[2]; i2 <= reps[2]; ++i2, ++c) {
        grid[3 * c] = (T)i0;
        grid[3 * c + 1] = (T)i1;
        grid[3 * c + 2] = (T)i2;
      }
  for (int64_t g = 0; g < G; ++g) {
    const T* a = cell + 9 * g;
    for (int64_t k = 0; k < nc; ++k)
      for (int j = 0; j < 3; ++j) {
        T v = rnd<T>(grid[3 * k] * a[j]);
        v = rnd<T>(v + rnd<T>(grid[3 * k + 1] * a[3 + j]));
        v = rnd<T>(v + rnd<T>(grid[3 * k + 2] * a[6 + j]));
        offs[(g * nc + k) * 3 + j] = v;
      }
    CellGeom<T> cg;
    cell_geom<T>(a, cg);
    for (int ax = 0; ax < 3; ++ax) {
      for (int j = 0; j < 3; ++j) recip[9 * g + 3 * ax + j] = cg.recip[ax][j];
      thr[3 * g + ax] = rnd<T>(rnd<T>((T)cutoff * cg.inv_min[ax]) + (T)1e-3);
    }
  }
}
}  // namespace xeq

extern "C" {

int xeq_version(void) { return 100; }
const char* xeq_last_error(void) { return xeq::g_err; }
int64_t xeq_launch_count(void) { return xeq::g_launches.load(std::memory_order_relaxed); }
int64_t xeq_launch_names(int64_t first, char* buf, int64_t cap) {
  const int64_t last = xeq::g_launches.load(std::memory_order_relaxed);
  if (first < 0 || first > last || last - first > xeq::LAUNCH_RING) return -1;
  int64_t need = 1;
  for (int64_t i = first; i < last; ++i) {
    const char* s = xeq::g_launch_ring[i & (xeq::LAUNCH_RING - 1)];
    need += (int64_t)strlen(s ? s : "?") + 1;
  }
  if (buf == nullptr || cap < need) return need;
  char* p = buf;
  for (int64_t i = first; i < last; ++i) {
    const char* s = xeq::g_launch_ring[i & (xeq::LAUNCH_RING - 1)];
    const size_t n = strlen(s ? s : "?");
    memcpy(p, s ? s : "?", n);
    p += n;
    *p++ = '\n';
  }
  *p = 0;
  return need;
}

int xeq_csr_rowptr(const int64_t* keys, int64_t n_keys, int64_t n_rows, int32_t* rowptr, void* stream) {
  XEQ_CHECK_ARG(n_keys >= 0 && n_rows >= 0 && n_keys < (1ll << 31), "xeq_csr_rowptr: bad sizes");
  int64_t n = n_rows + 1;
  hipLaunchKernelGGL(k_csr_rowptr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys,
                     n_keys, n_rows, rowptr);
  XEQ_CHECK_LAUNCH("xeq_csr_rowptr");
  return XEQ_OK;
}

static int sort_bits(int64_t n_rows) {
  int b = 1;
  while (b < 31 && (1ll << b) < n_rows) ++b;
  return b;
}
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int64_t xeq_csr_by_key_workspace(int64_t n_keys, int64_t n_rows) {
  if (n_keys < 0 || n_rows < 0 || n_keys >= (1ll << 31)) return -1;
  size_t temp = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                         (int32_t*)nullptr, (int)n_keys, 0, sort_bits(n_rows)) != hipSuccess)
    return -1;
  return (int64_t)(3 * align256((size_t)n_keys * 4) + align256(temp));
}

int xeq_csr_by_key(const int64_t* keys, int64_t n_keys, int64_t n_rows, void* workspace, int64_t workspace_bytes,
                   int32_t* rowptr, int32_t* perm, void* stream) {
  XEQ_CHECK_ARG(n_keys >= 0 && n_rows >= 0 && n_keys < (1ll << 31), "xeq_csr_by_key: bad sizes");
  const int64_t need = xeq_csr_by_key_workspace(n_keys, n_rows);
  XEQ_CHECK_ARG(workspace_bytes >= need, "xeq_csr_by_key: workspace of %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  const size_t seg = align256((size_t)n_keys * 4);
  int32_t* k_in = (int32_t*)w;
  int32_t* k_out = (int32_t*)(w + seg);
  int32_t* v_in = (int32_t*)(w + 2 * seg);
  void* temp = w + 3 * seg;
  size_t temp_bytes = (size_t)(workspace_bytes - 3 * (int64_t)seg);
  if (n_keys > 0) {
    hipLaunchKernelGGL(k_sort_prepare, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, keys, n_keys, k_in, v_in);
    hipError_t e_ = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int32_t*)k_in, k_out, (const int32_t*)v_in, perm,
                                                       (int)n_keys, 0, sort_bits(n_rows), st);
    XEQ_CHECK_ARG(e_ == hipSuccess, "xeq_csr_by_key: radix sort failed: %s", hipGetErrorString(e_));
  }
  hipLaunchKernelGGL(k_csr_rowptr32, dim3((unsigned)((n_rows + 256) / 256)), dim3(256), 0, st, (const int32_t*)k_out, n_keys,
                     n_rows, rowptr);
  XEQ_CHECK_LAUNCH("xeq_csr_by_key");
  return XEQ_OK;
}

int xeq_csr_by_key_bounded(const int64_t* keys, int64_t n_keys, int64_t n_rows, const int32_t* n_valid, void* workspace,
                           int64_t workspace_bytes, int32_t* rowptr, int32_t* perm, void* stream) {
  XEQ_CHECK_ARG(n_keys >= 0 && n_rows >= 0 && n_keys < (1ll << 31) && n_rows < (1ll << 31) - 1 && n_valid != nullptr, "xeq_csr_by_key_bounded: bad sizes");
  const int64_t need = xeq_csr_by_key_workspace(n_keys, n_rows + 1);
  XEQ_CHECK_ARG(workspace_bytes >= need, "xeq_csr_by_key_bounded: workspace of %lld bytes, need %lld (xeq_csr_by_key_workspace(n_keys, n_rows + 1))",
                (long long)workspace_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  const size_t seg = align256((size_t)n_keys * 4);
  int32_t* k_in = (int32_t*)w;
  int32_t* k_out = (int32_t*)(w + seg);
  int32_t* v_in = (int32_t*)(w + 2 * seg);
  void* temp = w + 3 * seg;
  size_t temp_bytes = (size_t)(workspace_bytes - 3 * (int64_t)seg);
  if (n_keys > 0) {
    hipLaunchKernelGGL(k_sort_prepare_bounded, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, keys, n_keys, n_valid,
                       (int32_t)n_rows, k_in, v_in);
    hipError_t e_ = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int32_t*)k_in, k_out, (const int32_t*)v_in, perm,
                                                       (int)n_keys, 0, sort_bits(n_rows + 1), st);
    XEQ_CHECK_ARG(e_ == hipSuccess, "xeq_csr_by_key_bounded: radix sort failed: %s", hipGetErrorString(e_));
  }
  hipLaunchKernelGGL(k_csr_rowptr32, dim3((unsigned)((n_rows + 256) / 256)), dim3(256), 0, st, (const int32_t*)k_out, n_keys,
                     n_rows, rowptr);
  XEQ_CHECK_LAUNCH("xeq_csr_by_key_bounded");
  return XEQ_OK;
}

int xeq_exclusive_scan_i32(const int32_t* counts, int64_t n, int32_t* out, void* stream) {
  XEQ_CHECK_ARG(n >= 0, "xeq_exclusive_scan_i32: n < 0");
  hipLaunchKernelGGL(k_exclusive_scan_i32, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, n, out);
  XEQ_CHECK_LAUNCH("xeq_exclusive_scan_i32");
  return XEQ_OK;
}

// grid-wide form: hipCUB's decoupled look-back scan over n + 1 items (item n reads as 0, so out[n] is the total)
namespace {
struct ScanIn {
  const int32_t* in;
  int64_t n;
  __host__ __device__ int32_t operator()(int64_t i) const { return i < n ? in[i] : 0; }
};
using ScanIter = hipcub::TransformInputIterator<int32_t, ScanIn, hipcub::CountingInputIterator<int64_t>>;
}  // namespace

int64_t xeq_exclusive_scan_i32_workspace(int64_t n) {
  if (n < 0 || n >= ((int64_t)1 << 31) - 1) return -1;
  size_t temp = 0;
  ScanIter it(hipcub::CountingInputIterator<int64_t>(0), ScanIn{nullptr, n});
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, temp, it, (int32_t*)nullptr, (int)(n + 1), (hipStream_t)0) != hipSuccess) return -1;
  return (int64_t)(temp + 16);
}

int xeq_exclusive_scan_i32_ws(const int32_t* counts, int64_t n, int32_t* out, void* workspace, int64_t workspace_bytes,
                              void* stream) {
  XEQ_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) - 1, "xeq_exclusive_scan_i32_ws: n = %lld out of range", (long long)n);
  const int64_t need = xeq_exclusive_scan_i32_workspace(n);
  XEQ_CHECK_ARG(out && workspace && need >= 0 && workspace_bytes >= need, "xeq_exclusive_scan_i32_ws: workspace of %lld bytes, %lld needed",
                (long long)workspace_bytes, (long long)need);
  XEQ_CHECK_ARG(n == 0 || counts, "xeq_exclusive_scan_i32_ws: null input");
  size_t temp = (size_t)workspace_bytes;
  ScanIter it(hipcub::CountingInputIterator<int64_t>(0), ScanIn{counts, n});
  const hipError_t e = hipcub::DeviceScan::ExclusiveSum(workspace, temp, it, out, (int)(n + 1), (hipStream_t)stream);
  if (e != hipSuccess) {
    xeq::set_error("xeq_exclusive_scan_i32_ws: %s", hipGetErrorString(e));
    return XEQ_ERR_LAUNCH;
  }
  return XEQ_OK;
}

int xeq_reverse_edge_map(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes, const int32_t* c_rowptr,
                         int32_t* rev, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && n_nodes >= 0, "xeq_reverse_edge_map: negative size");
  if (n_edges == 0) return XEQ_OK;
  hipLaunchKernelGGL(k_reverse_edge_map, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     edge_index, edge_index + n_edges, c_rowptr, n_edges, n_nodes, rev);
  XEQ_CHECK_LAUNCH("xeq_reverse_edge_map");
  return XEQ_OK;
}

int xeq_reverse_edge_map_pbc(int dtype, const int64_t* edge_index, const void* cell_offsets, int64_t n_edges, int64_t n_nodes,
                             const int32_t* c_rowptr, int32_t* rev, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && n_nodes >= 0, "xeq_reverse_edge_map_pbc: negative size");
  if (n_edges == 0) return XEQ_OK;
  XEQ_CHECK_ARG(edge_index && cell_offsets && c_rowptr && rev, "xeq_reverse_edge_map_pbc: NULL argument");
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_reverse_edge_map_pbc<T>), dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, (hipStream_t)stream, edge_index,
                       edge_index + n_edges, (const T*)cell_offsets, c_rowptr, n_edges, n_nodes, rev);
  });
  XEQ_CHECK_LAUNCH("xeq_reverse_edge_map_pbc");
  return XEQ_OK;
}

int xeq_radius_graph_count(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                           double cutoff, int32_t* deg, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0, "xeq_radius_graph_count: negative size");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_CHECK_ARG(n_graphs > 0, "xeq_radius_graph_count: nodes without graphs");
  XEQ_DISPATCH_FLOAT(dtype, {
    T rc = (T)cutoff;
    hipLaunchKernelGGL((k_radius_graph<T, false>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos, ptr, n_graphs, n_nodes, rc * rc, deg,
                       (const int32_t*)nullptr, (int64_t)0, (int64_t*)nullptr);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_count");
  return XEQ_OK;
}

int xeq_radius_graph_fill(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                          double cutoff, const int32_t* rowptr, int64_t n_edges, int64_t* edge_index,
                          void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_edges >= 0, "xeq_radius_graph_fill: negative size");
  if (n_nodes == 0 || n_edges == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    T rc = (T)cutoff;
    hipLaunchKernelGGL((k_radius_graph<T, true>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos, ptr, n_graphs, n_nodes, rc * rc, (int32_t*)nullptr,
                       rowptr, n_edges, edge_index);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_fill");
  return XEQ_OK;
}

/* Capacity form of the open-boundary list (runtime.GraphedStep): count[0] = raw[n_nodes], the true edge count, and rowptr = raw when the
 * list fits the capacity, ALL ZEROS (an empty list) when it does not.  A list cut at the capacity would no longer be symmetric, and the
 * symmetric shortcuts downstream (reverse-edge map, mirror walk) index by the reverse edge; an empty list is safe for every kernel.
 * The caller compares count with the capacity when it reads the results (GraphedStep.overflowed).  One launch, out of place. */
int xeq_rowptr_guard(const int32_t* raw, int64_t n_nodes, int64_t capacity, int32_t* rowptr, int32_t* count, void* stream) {
  XEQ_CHECK_ARG(raw && rowptr && raw != rowptr && count && n_nodes >= 0 && capacity >= 0 && capacity < ((int64_t)1 << 31),
                "xeq_rowptr_guard: bad arguments");
  hipLaunchKernelGGL(xeq::k_rowptr_guard, dim3((unsigned)((n_nodes + 256) / 256)), dim3(256), 0, (hipStream_t)stream, raw, n_nodes,
                     (int32_t)capacity, rowptr, count);
  XEQ_CHECK_LAUNCH("xeq_rowptr_guard");
  return XEQ_OK;
}

/* rowptr[0 .. n] = exclusive prefix sums of deg[0 .. n) with xeq_rowptr_guard's rule applied (capacity < 0: no guard), count[0]
 * (optional) = the true total, running_total[0] (optional, int64) += the true total (a benchmark's device-side edge counter, kept
 * inside the captured step): ONE launch by one workgroup for n <= 36 864 (the scan + guard pair is three); XEQ_ERR_UNSUPPORTED
 * above, where the caller keeps the grid-wide scan. */
int xeq_rowptr_from_degrees(const int32_t* deg, int64_t n_nodes, int64_t capacity, int32_t* rowptr, int32_t* count, int64_t* running_total,
                            void* stream) {
  XEQ_CHECK_ARG((deg || n_nodes == 0) && rowptr && n_nodes >= 0 && capacity < ((int64_t)1 << 31), "xeq_rowptr_from_degrees: bad arguments");
  if (n_nodes > SCAN_WG_MAX_ITEMS) {
    xeq::set_error("xeq_rowptr_from_degrees: %lld nodes (the one-workgroup form takes <= %lld)", (long long)n_nodes, (long long)SCAN_WG_MAX_ITEMS);
    return XEQ_ERR_UNSUPPORTED;
  }
  static const bool attr_ok = hipFuncSetAttribute((const void*)xeq::k_rowptr_from_degrees, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)wg_scan_lds_bytes(SCAN_WG_MAX_ITEMS)) == hipSuccess;
  XEQ_CHECK_ARG(attr_ok, "xeq_rowptr_from_degrees: cannot reserve %zu bytes of LDS", wg_scan_lds_bytes(SCAN_WG_MAX_ITEMS));
  hipLaunchKernelGGL(xeq::k_rowptr_from_degrees, dim3(1), dim3(SCAN_WG_THREADS), wg_scan_lds_bytes(n_nodes), (hipStream_t)stream, deg, n_nodes,
                     capacity, rowptr, count, running_total);
  XEQ_CHECK_LAUNCH("xeq_rowptr_from_degrees");
  return XEQ_OK;
}
int64_t xeq_rowptr_from_degrees_max(void) { return SCAN_WG_MAX_ITEMS; }

int xeq_pbc_image_counts(int dtype, const void* cell_host, int64_t n_graphs, const int32_t pbc[3], double cutoff, int32_t reps[3]) {
  XEQ_CHECK_ARG(n_graphs >= 1 && cell_host != nullptr && cutoff > 0, "xeq_pbc_image_counts: bad arguments");
  if (dtype == XEQ_F32) xeq::image_counts<float>((const float*)cell_host, n_graphs, pbc, cutoff, reps);
  else if (dtype == XEQ_F64) xeq::image_counts<double>((const double*)cell_host, n_graphs, pbc, cutoff, reps);
  else XEQ_CHECK_ARG(false, "xeq_pbc_image_counts: dtype %d", dtype);
  XEQ_CHECK_ARG(reps[0] <= 64 && reps[1] <= 64 && reps[2] <= 64, "xeq_pbc_image_counts: %d x %d x %d images per axis (degenerate cell?)",
                (int)reps[0], (int)reps[1], (int)reps[2]);
  return XEQ_OK;
}

int xeq_pbc_tables_host(int dtype, const void* cell_host, int64_t n_graphs, const int32_t reps[3], double cutoff, void* out_host,
                        int64_t out_count) {
  XEQ_CHECK_ARG(n_graphs >= 1 && cell_host != nullptr && out_host != nullptr && reps[0] >= 0 && reps[1] >= 0 && reps[2] >= 0,
                "xeq_pbc_tables_host: bad arguments");
  const int64_t nc = (int64_t)(2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1);
  XEQ_CHECK_ARG(out_count == (3 + 3 * n_graphs) * nc + 12 * n_graphs, "xeq_pbc_tables_host: out[] holds %lld values, need %lld",
                (long long)out_count, (long long)((3 + 3 * n_graphs) * nc + 12 * n_graphs));
  if (dtype == XEQ_F32) xeq::tables_host<float>((const float*)cell_host, n_graphs, reps, cutoff, (float*)out_host);
  else if (dtype == XEQ_F64) xeq::tables_host<double>((const double*)cell_host, n_graphs, reps, cutoff, (double*)out_host);
  else XEQ_CHECK_ARG(false, "xeq_pbc_tables_host: dtype %d", dtype);
  return XEQ_OK;
}

int xeq_pbc_wrap(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, const void* cell,
                 const void* cell_inv, const int32_t pbc[3], void* pos_wrap, void* shift, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0, "xeq_pbc_wrap: negative size");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_CHECK_ARG(n_graphs > 0, "xeq_pbc_wrap: nodes without graphs");
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_pbc_wrap<T>), dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T*)pos, ptr,
                       n_graphs, n_nodes, (const T*)cell, (const T*)cell_inv, (int)pbc[0], (int)pbc[1], (int)pbc[2], (T*)pos_wrap,
                       (T*)shift);
  });
  XEQ_CHECK_LAUNCH("xeq_pbc_wrap");
  return XEQ_OK;
}

int xeq_radius_graph_pbc_count(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                               int64_t n_nodes, const void* img, int64_t n_cells, double cutoff, int32_t* deg,
                               void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_cells > 0, "xeq_radius_graph_pbc_count: bad sizes");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radius_graph_pbc<T, false>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)img,
                       (const T*)nullptr, (const T*)nullptr, n_cells, (T)cutoff, deg, (const int32_t*)nullptr,
                       (int64_t)0, (int64_t*)nullptr, (T*)nullptr);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_count");
  return XEQ_OK;
}

int xeq_radius_graph_pbc_fill(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                              int64_t n_nodes, const void* img, const void* cells, const void* shift,
                              int64_t n_cells, double cutoff, const int32_t* rowptr, int64_t n_edges,
                              int64_t* edge_index, void* cell_offsets, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_cells > 0 && n_edges >= 0, "xeq_radius_graph_pbc_fill: bad sizes");
  if (n_nodes == 0 || n_edges == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radius_graph_pbc<T, true>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)img,
                       (const T*)cells, (const T*)shift, n_cells, (T)cutoff, (int32_t*)nullptr, rowptr, n_edges,
                       edge_index, (T*)cell_offsets);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_fill");
  return XEQ_OK;
}

int xeq_radius_graph_pbc_count_pruned(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                                      int64_t n_nodes, const void* img, int64_t n_cells, double cutoff, const void* recip,
                                      const void* thr, const int32_t reps[3], int32_t* deg, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_cells > 0, "xeq_radius_graph_pbc_count_pruned: bad sizes");
  XEQ_CHECK_ARG(reps[0] >= 0 && reps[1] >= 0 && reps[2] >= 0 &&
                    (int64_t)(2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1) == n_cells,
                "xeq_radius_graph_pbc_count_pruned: image counts (%d,%d,%d) do not match n_cells %lld", reps[0], reps[1],
                reps[2], (long long)n_cells);
  if (n_nodes == 0) return XEQ_OK;
  PbcPrune pr{{reps[0], reps[1], reps[2]}};
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radius_graph_pbc_img<T, false>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)img, (const T*)nullptr,
                       (const T*)nullptr, n_cells, (T)cutoff, (const T*)recip, (const T*)thr, pr, deg,
                       (const int32_t*)nullptr, (int64_t)0, (int64_t*)nullptr, (T*)nullptr);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_count_pruned");
  return XEQ_OK;
}

int xeq_radius_graph_pbc_fill_pruned(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                                     int64_t n_nodes, const void* img, const void* cells, const void* shift, int64_t n_cells,
                                     double cutoff, const void* recip, const void* thr, const int32_t reps[3],
                                     const int32_t* rowptr, int64_t n_edges, int64_t* edge_index, void* cell_offsets,
                                     void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_cells > 0 && n_edges >= 0, "xeq_radius_graph_pbc_fill_pruned: bad sizes");
  XEQ_CHECK_ARG((int64_t)(2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1) == n_cells,
                "xeq_radius_graph_pbc_fill_pruned: image counts do not match n_cells");
  if (n_nodes == 0 || n_edges == 0) return XEQ_OK;
  PbcPrune pr{{reps[0], reps[1], reps[2]}};
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radius_graph_pbc_img<T, true>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)img, (const T*)cells,
                       (const T*)shift, n_cells, (T)cutoff, (const T*)recip, (const T*)thr, pr, (int32_t*)nullptr, rowptr,
                       n_edges, edge_index, (T*)cell_offsets);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_fill_pruned");
  return XEQ_OK;
}

/* ---- cell-list form (graphs of many atoms): see k_radius_graph_pbc_cl */
int xeq_radius_graph_pbc_bin_ids(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                                 const void* recip, const int32_t* nbins, const int32_t* bin_base, int64_t* keys, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0, "xeq_radius_graph_pbc_bin_ids: bad sizes");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_pbc_bin_ids<T>), dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)recip, nbins, bin_base, keys);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_bin_ids");
  return XEQ_OK;
}

int xeq_radius_graph_pbc_count_cl(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                                  const void* img, int64_t n_cells, double cutoff, const void* recip, const void* thr,
                                  const int32_t reps[3], const int32_t* nbins, const int32_t* bin_base,
                                  const int32_t* bin_start, const int32_t* bin_atom, int32_t* deg, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_cells > 0, "xeq_radius_graph_pbc_count_cl: bad sizes");
  XEQ_CHECK_ARG((int64_t)(2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1) == n_cells,
                "xeq_radius_graph_pbc_count_cl: image counts do not match n_cells");
  if (n_nodes == 0) return XEQ_OK;
  PbcPrune pr{{reps[0], reps[1], reps[2]}};
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radius_graph_pbc_cl<T, false>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)img, n_cells, (T)cutoff,
                       (const T*)recip, (const T*)thr, pr, nbins, bin_base, bin_start, bin_atom, deg,
                       (const int32_t*)nullptr, (int64_t*)nullptr);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_count_cl");
  return XEQ_OK;
}

int xeq_radius_graph_pbc_fill_cl(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                                 const void* img, const void* cells, const void* shift, int64_t n_cells, double cutoff,
                                 const void* recip, const void* thr, const int32_t reps[3], const int32_t* nbins,
                                 const int32_t* bin_base, const int32_t* bin_start, const int32_t* bin_atom,
                                 const int32_t* rowptr, int64_t n_edges, int64_t* tmp_keys, int64_t* edge_index,
                                 void* cell_offsets, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_cells > 0 && n_edges >= 0, "xeq_radius_graph_pbc_fill_cl: bad sizes");
  XEQ_CHECK_ARG((int64_t)(2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1) == n_cells,
                "xeq_radius_graph_pbc_fill_cl: image counts do not match n_cells");
  if (n_nodes == 0 || n_edges == 0) return XEQ_OK;
  PbcPrune pr{{reps[0], reps[1], reps[2]}};
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_radius_graph_pbc_cl<T, true>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)pos_wrap, ptr, n_graphs, n_nodes, (const T*)img, n_cells, (T)cutoff,
                       (const T*)recip, (const T*)thr, pr, nbins, bin_base, bin_start, bin_atom, (int32_t*)nullptr, rowptr,
                       tmp_keys);
    hipLaunchKernelGGL((k_pbc_sort_segment<T>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const int64_t*)tmp_keys, rowptr, ptr, n_graphs, n_nodes, n_cells, (const T*)cells, (const T*)shift,
                       n_edges, edge_index, (T*)cell_offsets);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_pbc_fill_cl");
  return XEQ_OK;
}

/* ---- open-boundary cell list: see k_radius_graph_cl */
int xeq_radius_graph_bin_ids(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, const void* lo,
                             const void* inv_w, const int32_t* nbins, const int32_t* bin_base, int64_t* keys, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0, "xeq_radius_graph_bin_ids: bad sizes");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    hipLaunchKernelGGL((k_box_bin_ids<T>), dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)pos, ptr, n_graphs, n_nodes, (const T*)lo, (const T*)inv_w, nbins, bin_base, keys);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_bin_ids");
  return XEQ_OK;
}

int xeq_radius_graph_count_cl(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, double cutoff,
                              const void* lo, const void* inv_w, const int32_t* nbins, const int32_t* bin_base,
                              const int32_t* bin_start, const int32_t* bin_atom, int32_t* deg, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0, "xeq_radius_graph_count_cl: bad sizes");
  if (n_nodes == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    T rc = (T)cutoff;
    hipLaunchKernelGGL((k_radius_graph_cl<T, false>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)pos, ptr, n_graphs, n_nodes, rc * rc, (const T*)lo, (const T*)inv_w, nbins, bin_base, bin_start,
                       bin_atom, deg, (const int32_t*)nullptr, (int64_t*)nullptr);
  });
  XEQ_CHECK_LAUNCH("xeq_radius_graph_count_cl");
  return XEQ_OK;
}

int xeq_radius_graph_fill_cl(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, double cutoff,
                             const void* lo, const void* inv_w, const int32_t* nbins, const int32_t* bin_base,
                             const int32_t* bin_start, const int32_t* bin_atom, const int32_t* rowptr, int64_t n_edges,
                             int64_t* tmp_keys, int64_t* edge_index, void* stream) {
  XEQ_CHECK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_edges >= 0, "xeq_radius_graph_fill_cl: bad sizes");
  if (n_nodes == 0 || n_edges == 0) return XEQ_OK;
  XEQ_DISPATCH_FLOAT(dtype, {
    T rc = (T)cutoff;
    hipLaunchKernelGGL((k_radius_graph_cl<T, true>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)pos, ptr, n_graphs, n_nodes, rc * rc, (const T*)lo, (const T*)inv_w, nbins, bin_base, bin_start,
                       bin_atom, (int32_t*)nullptr, rowptr, tmp_keys);
  });
  hipLaunchKernelGGL(k_sort_segment_plain, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const int64_t*)tmp_keys, rowptr, n_nodes, n_edges, edge_index);
  XEQ_CHECK_LAUNCH("xeq_radius_graph_fill_cl");
  return XEQ_OK;
}

}  // extern "C"
