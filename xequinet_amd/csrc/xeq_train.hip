// Training-pass kernels of the message block on the matrix cores: the parameter gradients of the radial filter,
//   dL/dW_rbf[c, k], dL/db_rbf[c]   (rbf_lin, nn/xpainn.py:117,140)      dL/dfreq_k  or  dL/dmean_k, dL/dstd_k   (nn/rbf.py:143-146, :121-125)
// which the reference gets from autograd through the materialised filter[E, 576] (nn/basic.py:154-155, utils/trainer.py:295-302).
// Same contraction as k_message_param_grad (xeq_message.hip, the f64 / any-shape form): with
//   G[e, c] = dL/dfilter_out[e, c] h[nbr(e), c]      and the per-edge row   T[e, :] = f(d_e) [rho_k | 1 | d rho_k/d p0_k | d rho_k/d p1_k],
//   P[c, :] = sum_e G[e, c] T[e, :]   is a [rows x E] x [E x 41] product whose LEFT operand is formed on the fly.
// Here (f32, multiplicities and node_dim in multiples of 32):
//   * xeq_param_basis writes T (and Y_1, Y_2 of the edge) once per training step: the three blocks share the geometry;
//   * a wave owns 32 filter rows of one role (gate_state / gate_edge / msg_s) and a contiguous run of edges in the graph's own
//     (center-sorted) order; lane (i, p) forms G for row i and the edges of parity p -- its A fragment of v_mfma_f32_32x32x2_f32 --
//     and takes T[e, j] / T[e, 32 + j] as the B fragments: two MFMAs per edge pair, 32 accumulator registers (the node-walk form
//     holds 41 running sums per lane);
//   * the loop is bound by the COUNT of vector-memory instructions (see the kernel): table rows and indices come in 16 edges at a
//     time with 16-byte loads through wave-private LDS, the center's gradient rows stay in registers while the center stays;
//   * the four waves of a workgroup take quarters of the workgroup's edge range and add their tiles up through LDS: one part per
//     workgroup, parts[part][row][64], summed by the caller in a fixed order.
#include "xeq_common.h"

namespace xeq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// T row: [f rho_k (B) | f | f d rho_k/d p0 (B) | f d rho_k/d p1 (B, gaussian only) | pad to 4 | Y_1 (3) Y_2 (5) | pad to 32]
struct PbLayout {
  int B, ncols, ycol, W;
};
__host__ __device__ inline PbLayout pb_layout(int rbf_kind, int B) {
  PbLayout p;
  p.B = B;
  p.ncols = (rbf_kind == XEQ_RBF_GAUSSIAN ? 3 : 2) * B + 1;
  p.ycol = (p.ncols + 3) & ~3;
  p.W = (p.ycol + 8 + 31) & ~31;   // whole 128-byte lines per row: a half-wave's 32 columns sit in ONE line (208-byte rows cost two)
  return p;
}

// 16 threads per edge, four columns (one 16-byte store) each: a thread per edge walked its 20 basis functions serially and wrote
// its 256-byte row alone (0.20 ms per step on QM9-1024)
__global__ void k_param_basis(const float* __restrict__ vec, int64_t E, RadialSpec rs, const float* __restrict__ p0,
                              const float* __restrict__ p1, float* __restrict__ tab) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const PbLayout L = pb_layout(rs.rbf_kind, rs.num_basis);
  const int per_edge = L.W / 4;
  const int64_t e = t / per_edge;
  if (e >= E) return;
  const int c0 = 4 * (int)(t - e * per_edge);
  const float rc = (float)rs.cutoff;
  const EdgeGeom<float> g = edge_geom<float>(vec[3 * e], vec[3 * e + 1], vec[3 * e + 2]);
  float f, df;
  envelope<float>(rs.cutoff_kind, g.d, rc, f, df);
  float y[8];
  if (c0 + 4 > L.ycol && c0 < L.ycol + 8) sph_harm_l12<float>(g, y, y + 3);
  float out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + q;
    float v = 0.f;
    if (c < L.ncols) {
      if (c == L.B) {
        v = f;
      } else {
        const int which = c < L.B ? 0 : (c <= 2 * L.B ? 1 : 2);          // rho_k | d rho_k / d p0 | d rho_k / d p1
        const int k = which == 0 ? c : (which == 1 ? c - L.B - 1 : c - 2 * L.B - 1);
        const float q0 = p0[k], q1 = p1 ? p1[k] : 0.f;
        float rho, drho, d0, d1;
        if (which == 0) {
          radial<float>(rs.rbf_kind, g.d, rc, q0, q1, rho, drho);
          v = f * rho;
        } else {
          radial_dparam<float>(rs.rbf_kind, g.d, rc, q0, q1, d0, d1);
          v = f * (which == 1 ? d0 : d1);
        }
      }
    } else if (c >= L.ycol && c < L.ycol + 8) {
      v = y[c - L.ycol];
    }
    out[q] = v;
  }
  *reinterpret_cast<float4*>(tab + e * L.W + c0) = make_float4(out[0], out[1], out[2], out[3]);
}

// x[n, D] in the e3nn layout -> the BT layout (xeq_node.hip: per l a row-major [N (2l+1), mul_l] matrix): the center rows dL/dx_out
// the gradient kernel gathers are then contiguous over the channels (in the e3nn layout a half-wave's 32 channels of one m are 12 or
// 20 bytes apart: 4-6 cache lines per load instead of one -- the loads, not the MFMAs, bound this kernel)
__global__ void k_to_bt(const float* __restrict__ x, int64_t N, Irreps ir, float* __restrict__ out) {
  const int D = ir.D();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const int64_t n = t / D;
  const int f = (int)(t - n * D);
  int l, up, m;
  int64_t base;
  if (f < ir.mul[0]) { l = 0; up = f; m = 0; base = 0; }
  else if (f < ir.mul[0] + 3 * ir.mul[1]) { const int r = f - ir.mul[0]; l = 1; up = r / 3; m = r % 3; base = ir.mul[0]; }
  else { const int r = f - ir.mul[0] - 3 * ir.mul[1]; l = 2; up = r / 5; m = r % 5; base = (int64_t)ir.mul[0] + 3 * ir.mul[1]; }
  out[N * base + (n * (2 * l + 1) + m) * ir.mul[l] + up] = x[t];
}

// Affine-parameter gradients of the two norms of a block (nn.LayerNorm on the scalars, EquivariantLayerNorm on x: nn/o3layer.py:145-171)
// from the block inputs, the statistics the forward kernel kept (mean, rstd, mean of the 0e block, r = rsqrt(mean square norm)) and the
// gradients of the normalised features (g_shat rows with stride ld, g_xhat in the BT layout):
//   d ln_w[f] = sum_n g_shat[n, f] (s[n, f] - mean[n]) rstd[n]        d ln_b[f] = sum_n g_shat[n, f]
//   d eq_w[u] = sum_n r[n] sum_m g_xhat[n, u, m] (x[n, u, m] - [l = 0] mean0[n])        d eq_b[u] = sum_n g_xhat[n, u, 0]   (0e channels)
// parts[chunk][2 F + C + m0]; the caller sums the chunks.  As device tensor operations these were six reductions and a dozen
// elementwise launches per norm (six norms per training step).
__global__ void __launch_bounds__(256) k_norm_param_grad(const float* __restrict__ s, const float* __restrict__ x,
                                                         const float* __restrict__ stats, const float* __restrict__ g_shat, int64_t ld,
                                                         const float* __restrict__ g_xhat, int64_t N, int F, Irreps ir,
                                                         int64_t rows_per_chunk, float* __restrict__ parts) {
  const int C = ir.C(), D = ir.D(), m0 = ir.mul[0];
  const int W = 2 * F + C + m0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
  float* out = parts + (int64_t)blockIdx.x * W;
  for (int col = threadIdx.x; col < F + C; col += blockDim.x) {
    if (col < F) {
      float aw = 0.f, ab = 0.f;
      for (int64_t n = r0; n < r1; ++n) {
        const float g = g_shat[n * ld + col];
        aw += g * ((s[n * F + col] - stats[4 * n]) * stats[4 * n + 1]);
        ab += g;
      }
      out[col] = aw;
      out[F + col] = ab;
    } else {
      const int u = col - F;
      int l, off;
      ir.locate(u, l, off);
      const int nm = 2 * l + 1;
      const XAddr ga = xaddr(ir, N, u, 1);
      float aw = 0.f, ab = 0.f;
      for (int64_t n = r0; n < r1; ++n) {
        const float mean0 = l == 0 ? stats[4 * n + 2] : 0.f, r = stats[4 * n + 3];
        float acc = 0.f;
        for (int m = 0; m < nm; ++m) {
          const float g = g_xhat[ga.off + n * ga.node + (int64_t)m * ga.comp];
          acc += g * (x[n * D + off + m] - mean0);
          if (l == 0) ab += g;
        }
        aw += r * acc;
      }
      out[2 * F + u] = aw;
      if (u < m0) out[2 * F + C + u] = ab;
    }
  }
}

struct PgArgs {
  int64_t n_nodes, n_edges, edges_per_part;
  const int64_t* center;
  const int64_t* nbr;
  int F, C, D, H;
  Irreps ir;
  int xl, gl;   // layouts of xhat and of grad_x (XAddr: 0 e3nn, 1 BT)
  PbLayout L;
  int tiles_c, tiles_f, n_tiles;   // 32-row tiles: C / 32 per gate role, F / 32 for msg_s
  int n_parts;
  const int32_t* n_valid;          // optional: device-side edge count of a capacity-sized list (edges behind it are not walked)
};

constexpr int PG_CH = 16;        // edges a wave stages at a time
#ifndef XEQ_PG_STEPS
#define XEQ_PG_STEPS 2
#endif
constexpr int PG_STEPS = XEQ_PG_STEPS;   // edge pairs whose gathers are in flight together (measured 1 / 2 / 4 / 8: 859 / 719 / 871 / 901 us; 148 VGPRs at 2)
#ifdef XEQ_PG_WPE
#define XEQ_PG_OCC __attribute__((amdgpu_waves_per_eu(XEQ_PG_WPE, XEQ_PG_WPE)))
#else
#define XEQ_PG_OCC
#endif
constexpr int PG_LD = 96;        // floats per staged row: 64 used; odd and even edges land in different bank halves
#ifndef XEQ_PG_ABLATE            // development switch (scratch/bench_param_grad.py): bit 0 no center rows, 1 no xhat rows, 3 no h, 4 no MFMA
#define XEQ_PG_ABLATE 0
#endif

// What bounds this kernel is the NUMBER of vector-memory instructions, not their bytes: a dword-per-lane load costs the CU ~20-25
// cycles whatever its alignment (ablation, scratch/bench_param_grad.py: 8.3 loads per edge pair -> 0.97 ms without a single MFMA,
// 0.21 ms with the MFMAs and the index loads only).  So: the table rows and the edge indices of 16 edges come in with FOUR 16-byte
// loads per lane and go through (wave-private) LDS; per edge pair that leaves the row gathers -- h (1), the center's dL/ds_out or
// dL/dx_out (1-5: L1 hits, ~17 consecutive edges share a center) and, for gate_state tiles, xhat (1-5) -- all of a round requested
// together.
__global__ void __launch_bounds__(256) XEQ_PG_OCC k_message_param_grad_mc(PgArgs a, const float* __restrict__ tab, const float* __restrict__ h,
                                                               const float* __restrict__ xhat, const float* __restrict__ grad_s,
                                                               const float* __restrict__ grad_x, float* __restrict__ parts) {
  // staging rows of the four waves; reused for the cross-wave sum at the end (3 x 64 x 33 floats fit)
  constexpr int PG_ST = 4 * PG_CH * PG_LD > 3 * 64 * 33 ? 4 * PG_CH * PG_LD : 3 * 64 * 33;
  __shared__ __attribute__((aligned(16))) float sT[PG_ST];
  __shared__ int sC[4][PG_CH], sN[4][PG_CH];
  const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroup -> (tile, part): the tiles of a part all read the part's table rows, so they sit on ONE XCD (workgroups go to the XCDs
  // round-robin by their index) next to each other in dispatch order: the rows come from HBM once and hit that XCD's L2 afterwards.
  // (With the tiles spread over the XCDs 64 % of the L2 requests missed -- TCC_MISS / TCC_REQ -- and the kernel ran at the miss
  // parallelism of the L1s, TCP_PENDING_STALL 65 % of the time: 1.03 ms.)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = slot % a.n_tiles, part = (slot / a.n_tiles) * 8 + xcd;
  if (part >= a.n_parts) return;
  // tile -> role and first channel
  int role, cb;
  if (tile < a.tiles_c) { role = 0; cb = 32 * tile; }
  else if (tile < 2 * a.tiles_c) { role = 1; cb = 32 * (tile - a.tiles_c); }
  else { role = 2; cb = 32 * (tile - 2 * a.tiles_c); }
  const int ch = cb + i;                       // channel u (roles 0, 1) or scalar column (role 2)
  const int row = role * a.C + ch;             // filter row [gate_state C | gate_edge C | msg_s F]
  int l = 0, off = 0;
  if (role < 2) a.ir.locate(ch, l, off);       // the tile sits inside one l block (multiplicities in multiples of 32)
  const int nm = role < 2 ? 2 * l + 1 : 0;
  const int yoff = a.L.ycol + (l == 1 ? 0 : 3);   // Y_1 at ycol, Y_2 at ycol + 3 (l = 0: Y = 1, not read)
  const XAddr xa = xaddr(a.ir, a.n_nodes, role < 2 ? ch : 0, a.xl);
  const XAddr ga = xaddr(a.ir, a.n_nodes, role < 2 ? ch : 0, a.gl);
  float* sTw = sT + wave * (PG_CH * PG_LD);
  // the workgroup's edges in chunks of 16, dealt to the four waves in turn: the workgroup -- and the other tiles of the part, which
  // start with it -- moves through the range as one narrow window (what has to stay in L2)
  const int64_t w0 = (int64_t)part * a.edges_per_part;
  const int64_t n_live = a.n_valid ? min(a.n_edges, (int64_t)a.n_valid[0]) : a.n_edges;
  const int s1 = (int)max((int64_t)0, min(n_live, w0 + a.edges_per_part) - w0);   // edges of the part: positions below are relative to w0 (32-bit)
  const int s0 = min(s1, wave * PG_CH);
  const float4* tabp = reinterpret_cast<const float4*>(tab) + w0 * 16;
  const int64_t* ctrp = a.center + w0;
  const int64_t* nbrp = a.nbr + w0;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;

  // chunk loads: 16 rows x 16 float4 = 4 per lane (lane -> row idx >> 4, float4 idx & 15); lanes 0..15 / 16..31 the center / neighbor
  float4 pre[4];
  int pre_idx = 0;
  auto issue = [&](int base) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = lane + 64 * k;
      const int e = base + (idx >> 4);
      pre[k] = e < s1 ? tabp[e * 16 + (idx & 15)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int e = base + (lane & 15);
    pre_idx = (lane < 32 && e < s1) ? (int)(lane < 16 ? ctrp[e] : nbrp[e]) : 0;   // past the range: node 0, its G dropped
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = lane + 64 * k;
      *reinterpret_cast<float4*>(&sTw[(idx >> 4) * PG_LD + 4 * (idx & 15)]) = pre[k];
    }
    if (lane < 16) sC[wave][lane] = pre_idx;
    else if (lane < 32) sN[wave][lane - 16] = pre_idx;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  if (s0 < s1) issue(s0);
  for (int base = s0; base < s1; base += 4 * PG_CH) {
    commit();
    if (base + 4 * PG_CH < s1) issue(base + 4 * PG_CH);     // in flight under this chunk's gathers and MFMAs
#pragma unroll
    for (int half = 0; half < 8 / PG_STEPS; ++half) {
      // every row gather of the round is requested before the first is used: no branch and no use in between (a lane-held copy of
      // the center's rows behind an `if (center changed)` put a wait into every edge pair: four latencies per round instead of one)
      float hv[PG_STEPS], xo[PG_STEPS][5], gc[PG_STEPS][5];
#pragma unroll
      for (int u = 0; u < PG_STEPS; ++u) {
        const int j = 2 * PG_STEPS * half + 2 * u + kh;
        const int64_t c = sC[wave][j], n = sN[wave][j];
        hv[u] = (XEQ_PG_ABLATE & 8) ? 1.f : h[n * a.H + row];
        if (role == 2) {
          gc[u][0] = (XEQ_PG_ABLATE & 1) ? 1.f : grad_s[c * a.F + ch];
        } else {
          const float* gx = grad_x + ga.off + c * ga.node;
          const float* xn = xhat + xa.off + n * xa.node;
#pragma unroll
          for (int m = 0; m < 5; ++m) {
            if (m < nm) {
              gc[u][m] = (XEQ_PG_ABLATE & 1) ? 1.f : gx[(int64_t)m * ga.comp];
              if (role == 0) xo[u][m] = (XEQ_PG_ABLATE & 2) ? 1.f : xn[(int64_t)m * xa.comp];
            }
          }
        }
      }
      float G[PG_STEPS], b0[PG_STEPS], b1[PG_STEPS];
#pragma unroll
      for (int u = 0; u < PG_STEPS; ++u) {
        const int j = 2 * PG_STEPS * half + 2 * u + kh;
        const bool ok = base + j < s1;
        float dg;
        if (role == 2) {
          dg = gc[u][0];
        } else {
          dg = 0.f;
#pragma unroll
          for (int m = 0; m < 5; ++m)
            if (m < nm) dg += (role == 0 ? xo[u][m] : (l == 0 ? 1.f : sTw[j * PG_LD + yoff + m])) * gc[u][m];
        }
        G[u] = ok ? hv[u] * dg : 0.f;
        b0[u] = sTw[j * PG_LD + i];            // columns past ncols are the row's zero padding
        b1[u] = sTw[j * PG_LD + 32 + i];
      }
#pragma unroll
      for (int u = 0; u < PG_STEPS; ++u) {
        if (XEQ_PG_ABLATE & 16) {
          acc0[0] += G[u] * b0[u];
          acc1[0] += G[u] * b1[u];
          continue;
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(G[u], b0[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(G[u], b1[u], acc1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();   // (LDS operations of a wave complete in order: the next commit's stores follow these reads)
  }
  // waves 1..3 hand their tiles to wave 0 (fixed order 0 + 1 + 2 + 3), which writes the part
  __syncthreads();                     // every wave is done with its staging rows
  float (*red)[64][33] = reinterpret_cast<float (*)[64][33]>(sT);
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[wave - 1][lane][r] = acc0[r];
      red[wave - 1][lane][16 + r] = acc1[r];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] += red[w][lane][r];
        acc1[r] += red[w][lane][16 + r];
      }
    float* out = parts + ((int64_t)part * a.H + role * a.C + cb) * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;      // D layout: row of register r
      out[rr * 64 + i] = acc0[r];
      out[rr * 64 + 32 + i] = acc1[r];
    }
  }
}

}  // namespace xeq

using namespace xeq;

extern "C" {

int xeq_message_param_grad_mc_supported(int dtype, int rbf_kind, int num_basis, int node_dim, const int32_t mul[3]) {
  if (dtype != XEQ_F32 || num_basis < 1 || node_dim < 32 || node_dim % 32) return 0;
  if (rbf_kind != XEQ_RBF_BESSEL && rbf_kind != XEQ_RBF_GAUSSIAN) return 0;
  if (pb_layout(rbf_kind, num_basis).W != 64) return 0;   // table and harmonics in one 256-byte row
  int c = 0;
  for (int l = 0; l < 3; ++l) {
    if (mul[l] < 0 || mul[l] % 32) return 0;
    c += mul[l];
  }
  return c >= 32 ? 1 : 0;
}

int xeq_param_basis_width(int rbf_kind, int num_basis) { return pb_layout(rbf_kind, num_basis).W; }

/* parts of xeq_message_param_grad_mc: workgroups per 32-row tile, ~8 waves per SIMD over all tiles, at least 64 edges per workgroup */
int xeq_message_param_grad_mc_parts(int64_t n_edges, int node_dim, const int32_t mul[3]) {
  const int64_t tiles = 2 * ((mul[0] + mul[1] + mul[2]) / 32) + node_dim / 32;
  int64_t p = (2048 + tiles - 1) / tiles, pmax = (n_edges + 63) / 64;
  if (p > pmax) p = pmax;
  return (int)(p < 1 ? 1 : p);
}

int xeq_param_basis(const void* vec, int64_t n_edges, int rbf_kind, int cutoff_kind, int num_basis, double cutoff, const void* p0,
                    const void* p1, void* tab, void* stream) {
  XEQ_CHECK_ARG(n_edges >= 0 && num_basis >= 1 && cutoff > 0, "xeq_param_basis: bad sizes");
  XEQ_CHECK_ARG(rbf_kind == XEQ_RBF_BESSEL || (rbf_kind == XEQ_RBF_GAUSSIAN && p1 != nullptr), "xeq_param_basis: rbf kernel %d (gaussian needs std)", rbf_kind);
  XEQ_CHECK_ARG(cutoff_kind == XEQ_CUTOFF_COSINE || cutoff_kind == XEQ_CUTOFF_POLYNOMIAL, "xeq_param_basis: cutoff function %d is not implemented", cutoff_kind);
  if (n_edges == 0) return XEQ_OK;
  const int64_t threads = n_edges * (pb_layout(rbf_kind, num_basis).W / 4);
  hipLaunchKernelGGL(k_param_basis, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)vec, n_edges,
                     RadialSpec{rbf_kind, cutoff_kind, num_basis, cutoff}, (const float*)p0, (const float*)p1, (float*)tab);
  XEQ_CHECK_LAUNCH("xeq_param_basis");
  return XEQ_OK;
}

int xeq_norm_param_grad_chunks(int64_t n_nodes) {
  // a thread walks its chunk's rows serially (one dependent load round trip per row): short chunks -- 16 rows -- keep that walk at
  // ~25 us; 73 rows took 128 us per launch
  const int64_t c = (n_nodes + 15) / 16;
  return (int)(c < 1 ? 1 : (c > 4096 ? 4096 : c));
}

int xeq_norm_param_grad(const void* s, const void* x, const void* stats, const void* g_shat, int64_t ld_gs, const void* g_xhat_bt,
                        int64_t n_nodes, int node_dim, const int32_t mul[3], int n_chunks, void* parts, void* stream) {
  XEQ_CHECK_ARG(n_nodes >= 0 && node_dim >= 1 && mul[0] >= 0 && mul[1] >= 0 && mul[2] >= 0 && ld_gs >= node_dim, "xeq_norm_param_grad: bad sizes");
  XEQ_CHECK_ARG(n_chunks == xeq_norm_param_grad_chunks(n_nodes), "xeq_norm_param_grad: parts must hold xeq_norm_param_grad_chunks(n) = %d rows, got %d",
                xeq_norm_param_grad_chunks(n_nodes), n_chunks);
  Irreps ir{{mul[0], mul[1], mul[2]}};
  const int64_t rows = (n_nodes + n_chunks - 1) / n_chunks;
  hipLaunchKernelGGL(k_norm_param_grad, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const float*)s, (const float*)x,
                     (const float*)stats, (const float*)g_shat, ld_gs, (const float*)g_xhat_bt, n_nodes, node_dim, ir, rows, (float*)parts);
  XEQ_CHECK_LAUNCH("xeq_norm_param_grad");
  return XEQ_OK;
}

int xeq_to_bt(const void* x, int64_t n_nodes, const int32_t mul[3], void* out, void* stream) {
  XEQ_CHECK_ARG(n_nodes >= 0 && mul[0] >= 0 && mul[1] >= 0 && mul[2] >= 0, "xeq_to_bt: bad sizes");
  Irreps ir{{mul[0], mul[1], mul[2]}};
  const int64_t total = n_nodes * ir.D();
  if (total == 0) return XEQ_OK;
  hipLaunchKernelGGL(k_to_bt, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, n_nodes, ir, (float*)out);
  XEQ_CHECK_LAUNCH("xeq_to_bt");
  return XEQ_OK;
}

int xeq_message_param_grad_mc(int64_t n_nodes, int64_t n_edges, const int64_t* center, const int64_t* nbr, const void* tab, const void* h,
                              const void* xhat, const void* grad_s, const void* grad_x, int rbf_kind, int num_basis, int node_dim,
                              const int32_t mul[3], int xhat_layout, int grad_x_layout, const int32_t* n_valid, int n_parts, void* parts,
                              void* stream) {
  XEQ_CHECK_ARG(xeq_message_param_grad_mc_supported(XEQ_F32, rbf_kind, num_basis, node_dim, mul),
                "xeq_message_param_grad_mc: f32, node_dim and multiplicities in multiples of 32, table and harmonics within 64 columns (num_basis %d)", num_basis);
  XEQ_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_parts == xeq_message_param_grad_mc_parts(n_edges, node_dim, mul),
                "xeq_message_param_grad_mc: parts must hold xeq_message_param_grad_mc_parts() = %d blocks, got %d",
                xeq_message_param_grad_mc_parts(n_edges, node_dim, mul), n_parts);
  PgArgs a{};
  a.n_nodes = n_nodes;
  a.n_edges = n_edges;
  a.center = center;
  a.nbr = nbr;
  for (int l = 0; l < 3; ++l) a.ir.mul[l] = mul[l];
  a.F = node_dim;
  a.C = a.ir.C();
  a.D = a.ir.D();
  a.H = a.F + 2 * a.C;
  a.xl = xhat_layout & 1;
  a.gl = grad_x_layout & 1;
  a.L = pb_layout(rbf_kind, num_basis);
  a.tiles_c = a.C / 32;
  a.tiles_f = a.F / 32;
  a.n_tiles = 2 * a.tiles_c + a.tiles_f;
  a.n_parts = n_parts;
  a.n_valid = n_valid;
  a.edges_per_part = ((n_edges + n_parts - 1) / n_parts + 1) & ~(int64_t)1;
  hipLaunchKernelGGL(k_message_param_grad_mc, dim3((unsigned)(a.n_tiles * ((n_parts + 7) / 8) * 8)), dim3(256), 0, (hipStream_t)stream, a, (const float*)tab,
                     (const float*)h, (const float*)xhat, (const float*)grad_s, (const float*)grad_x, (float*)parts);
  XEQ_CHECK_LAUNCH("xeq_message_param_grad_mc");
  return XEQ_OK;
}

}  // extern "C"
