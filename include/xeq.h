/* libxeq_hip.so -- C ABI of the MI355X-native XPaiNN energy+force hot path.
 *
 * Drop-in boundary.  The reference (X1X1010/XequiNet) is pure
 * Python and has no FFI; its hot path bottoms out in third-party extension ops.
 * Each entry point below names the reference call site it replaces
 * (paths relative to /root/reference/xequinet/).  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    marked "host".  All buffers are owned by the caller (e.g. the PyTorch
 *    caching allocator); the library never allocates, frees or retains them.
 *  - Row-major contiguous tensors.  `dtype` selects f32/f64 for all floating
 *    tensors of a call.  Index tensors are int64 where the reference's are
 *    (edge_index, ptr), int32 for the library's own CSR arrays.
 *  - Every call enqueues on `stream` (a hipStream_t passed as void*) and
 *    returns without synchronising; re-entrant, no global mutable state
 *    besides the thread-local error string.
 *  - Return value: 0 = ok, otherwise an XEQ_ERR_* code; xeq_last_error()
 *    returns the message for the calling thread (the reference raises Python
 *    exceptions/asserts, e.g. nn/rbf.py:44-46, nn/o3layer.py:105-107).
 *  - Irreps are passed as mul[3] = channels of l = 0,1,2 (0 = absent), e3nn
 *    mul_ir layout; C = sum mul, D = mul0 + 3 mul1 + 5 mul2.
 */
#ifndef XEQ_H
#define XEQ_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { XEQ_F32 = 0, XEQ_F64 = 1 };
/* radial bases of nn/rbf.py: SphericalBesselj0 (p0 = freq), GaussianSmearing (p0 = mean, p1 = std), ExponentialBernstein (p0 =
 * softplus(_alpha) repeated num_basis times, p1 = the log binomials `logc`; nn/rbf.py:161-191), ExponentialNorm (p0 = beta, p1 = mu;
 * nn/rbf.py:194-207).  The last two: inference entry points (xeq_radial_fwd, xeq_edge_basis, xeq_edge_basis_wq, xeq_message_fwd / _bwd);
 * the parameter-gradient kernels of the native training pass take the first two only. */
enum { XEQ_RBF_BESSEL = 0, XEQ_RBF_GAUSSIAN = 1, XEQ_RBF_EXPBERN = 2, XEQ_RBF_EXPNORM = 3 };
enum { XEQ_CUTOFF_COSINE = 0, XEQ_CUTOFF_POLYNOMIAL = 1 };
enum {
  XEQ_OK = 0,
  XEQ_ERR_INVALID_ARGUMENT = 1,
  XEQ_ERR_LAUNCH = 2,
  XEQ_ERR_UNSUPPORTED = 3
};

int xeq_version(void);
const char* xeq_last_error(void);
/* Kernel launches this library has enqueued so far in the process (all threads, every entry point, whichever front called it).
 * Test infrastructure: tests/test_gpu_parity.py::_aten_only proves with it that the ATen-only evaluation of the reference's op
 * sequence -- a member of the fp32 error envelope -- launched no kernel of this library. */
int64_t xeq_launch_count(void);
/* The entry-point names of launches number first .. xeq_launch_count() - 1, one per line, into buf (NUL-terminated).  -> the bytes needed
 * (call with buf = NULL to size it); -1 when `first` is out of range or older than the last 8 192 launches.  Test infrastructure:
 * tests/test_gpu_interface.py compares the launch SEQUENCES of the two fronts (Python modules, xeq::xpainn_eval) with it. */
int64_t xeq_launch_names(int64_t first, char* buf, int64_t cap);

/* ------------------------------------------------------------------ graph */

/* rowptr[i] = first position p with keys[p] >= i, i in [0, n_rows]  (keys sorted
 * ascending).  Turns a destination-sorted edge_index row into CSR. */
int xeq_csr_rowptr(const int64_t* keys, int64_t n_keys, int64_t n_rows, int32_t* rowptr, void* stream);

/* CSR view of an UNSORTED index row (edge_index[1] for the reverse pass, edge_index[0] of a user-built edge list):
 * perm[E] = stable order of the edges by key (a radix sort: ties keep edge order, deterministic), rowptr[n_rows+1] over
 * the sorted keys.  Replaces the stable torch.sort + searchsorted plumbing with one call; `workspace` is a device
 * scratch buffer of at least xeq_csr_by_key_workspace(n_keys, n_rows) bytes (-1: sizes out of range). */
int64_t xeq_csr_by_key_workspace(int64_t n_keys, int64_t n_rows);
int xeq_csr_by_key(const int64_t* keys, int64_t n_keys, int64_t n_rows, void* workspace, int64_t workspace_bytes,
                   int32_t* rowptr, int32_t* perm, void* stream);
/* The same for a CAPACITY-sized key array whose first *n_valid entries are edges (n_valid: device pointer, e.g. the last entry of the
 * center row pointer of a list written without reading its size back): the slots behind them sort to the end as row n_rows, so
 * rowptr[n_rows] = *n_valid and no walk reaches them.  Workspace: xeq_csr_by_key_workspace(n_keys, n_rows + 1).  What lets a
 * periodic neighbour list and its reverse view sit inside one captured HIP graph (runtime.GraphedStepPBC). */
int xeq_csr_by_key_bounded(const int64_t* keys, int64_t n_keys, int64_t n_rows, const int32_t* n_valid, void* workspace,
                           int64_t workspace_bytes, int32_t* rowptr, int32_t* perm, void* stream);

/* Exclusive prefix sum of int32 counts[n] into out[n+1] (out[n] = total). */
int xeq_exclusive_scan_i32(const int32_t* counts, int64_t n, int32_t* out, void* stream);
/* The same scan grid-wide (decoupled look-back), for large n; `workspace`: device scratch of at least
 * xeq_exclusive_scan_i32_workspace(n) bytes (-1: n out of range).  The form above runs in ONE workgroup and needs none. */
int64_t xeq_exclusive_scan_i32_workspace(int64_t n);
int xeq_exclusive_scan_i32_ws(const int32_t* counts, int64_t n, int32_t* out, void* workspace, int64_t workspace_bytes,
                              void* stream);

/* Replaces torch_cluster.radius_graph at data/transform.py:58-64 (non-PBC):
 * same-graph pairs with d^2 < r^2 (strict), no self loops, unlimited neighbours.
 * Phase 1 writes deg[N]; the caller scans it (xeq_exclusive_scan_i32), reads
 * E = rowptr[N] back, allocates edge_index[2,E]; phase 2 fills it in canonical
 * order: sorted by center (row 0), then neighbor (row 1). */
/* Capacity form (round 3): the edge count need not reach the host.  Pass a CAPACITY as n_edges (edge_index [2, capacity], never
 * written past) and hand the same capacity to xeq_reverse_edge_map, xeq_edge_vectors_fwd and the wq message entry points: every
 * walk over the list is bounded by rowptr[n_nodes] on the device, slots behind it keep whatever valid node ids they held (zero-
 * initialise the buffer once).  This is what lets neighbour list + model run as ONE captured HIP graph for batches of changing
 * edge counts (xequinet_amd/runtime.py::GraphedStep); an open-boundary list never exceeds sum_g n_g (n_g - 1). */
int xeq_radius_graph_count(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                           double cutoff, int32_t* deg, void* stream);
int xeq_radius_graph_fill(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                          double cutoff, const int32_t* rowptr, int64_t n_edges, int64_t* edge_index, void* stream);

/* Cell-list form of the open-boundary search for graphs of many atoms (the sweep above is O(n_g^2) per graph): per
 * graph an axis-aligned bin grid over its bounding box, lo[G,3] / inv_w[G,3] (inverse bin width, width >= cutoff) /
 * nbins[G,3], bin_base[G+1] = running bin count.  xeq_radius_graph_bin_ids gives each atom's global bin id, the caller
 * sorts atoms by it (xeq_csr_by_key: bin_start, bin_atom); count/fill visit the 3x3x3 bin block of each center with the
 * same d^2 < r^2 arithmetic and rank the hits per center: the same canonical edge_index, bit for bit. */
int xeq_radius_graph_bin_ids(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, const void* lo,
                             const void* inv_w, const int32_t* nbins, const int32_t* bin_base, int64_t* keys, void* stream);
int xeq_radius_graph_count_cl(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, double cutoff,
                              const void* lo, const void* inv_w, const int32_t* nbins, const int32_t* bin_base,
                              const int32_t* bin_start, const int32_t* bin_atom, int32_t* deg, void* stream);
int xeq_radius_graph_fill_cl(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, double cutoff,
                             const void* lo, const void* inv_w, const int32_t* nbins, const int32_t* bin_base,
                             const int32_t* bin_start, const int32_t* bin_atom, const int32_t* rowptr, int64_t n_edges,
                             int64_t* tmp_keys, int64_t* edge_index, void* stream);

/* Everything the periodic search derives from the cells alone (HOST function: host pointers in and out, no device work).
 * Replaces the host-side preparation of data/radius_graph.py: images per axis reps[a] = max_g ceil(cutoff |a_j x a_k| / V)
 * on the periodic axes, 0 on the open ones (:61-89); the image table grid[n_cells, 3] in cartesian_prod order, first axis
 * slowest (:93-97); its Cartesian offsets offs[G, n_cells, 3] = grid cell[g] (:98-104); and, for the image-pruned kernels
 * below, recip[G, 3, 3] = rows a_j x a_k / V and thr[G, 3] = cutoff |recip_a| + 1e-3.  Arithmetic in the cells' own dtype
 * with every operation rounded once (no contraction), in the order written here, so that every caller -- the Python front
 * (data/radius_graph.py of this package) and the registered operator xeq::radius_graph_pbc -- hands the search kernels the
 * same bits.  Two phases: xeq_pbc_image_counts gives reps (n_cells = prod (2 reps + 1)); xeq_pbc_tables_host fills
 * out[] = grid | offs | recip | thr, (3 + 3 G) n_cells + 12 G values of `dtype`. */
int xeq_pbc_image_counts(int dtype, const void* cell_host, int64_t n_graphs, const int32_t pbc[3], double cutoff, int32_t reps[3]);
int xeq_pbc_tables_host(int dtype, const void* cell_host, int64_t n_graphs, const int32_t reps[3], double cutoff, void* out_host,
                        int64_t out_count);

/* wrap_positions (data/radius_graph.py:6-32) for every atom: fractional = pos cell_inv[g], shift = floor(fractional) on the periodic
 * axes (pbc[a] != 0), pos_wrap = (fractional - shift) cell[g].  cell / cell_inv [G, 3, 3] row-major (rows = lattice vectors). */
int xeq_pbc_wrap(int dtype, const void* pos, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes, const void* cell,
                 const void* cell_inv, const int32_t pbc[3], void* pos_wrap, void* shift, void* stream);
/* Replaces the cdist/nonzero search of radius_graph_pbc (data/radius_graph.py:118-126,
 * 162-181).  The caller supplies what the reference computes on the host side:
 * wrapped positions pos_wrap[N,3] (:111), per-graph image translation vectors
 * img[G, n_cells, 3] (:104) and the integer image table cells[n_cells,3] (:97),
 * per-atom wrap shift[N,3] (:113).  Pairs with 0.01 < D < cutoff are emitted
 * center-major, then by (neighbor * n_cells + cell) ascending, bit-identical to the
 * reference ordering; cell_offsets = cells[cell] + shift[center] - shift[neighbor]
 * (:186-190). */
int xeq_radius_graph_pbc_count(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                               int64_t n_nodes, const void* img, int64_t n_cells, double cutoff, int32_t* deg,
                               void* stream);
int xeq_radius_graph_pbc_fill(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                              int64_t n_nodes, const void* img, const void* cells, const void* shift,
                              int64_t n_cells, double cutoff, const int32_t* rowptr, int64_t n_edges,
                              int64_t* edge_index, void* cell_offsets, void* stream);

/* The same search with image pruning (default of the Python front): per (center, neighbor) pair only the image
 * offsets n with |f_a - n_a| <= thr_a on every periodic axis are evaluated, f = (pos_wrap[c] - pos_wrap[n]) . recip^T --
 * a necessary condition for the pair to be within the cutoff, so the edge set and its order are bit-identical to the
 * exhaustive form above (the survivors are evaluated with the same arithmetic).  recip[G,3,3]: rows a_j x a_k / V of
 * each cell (what data/radius_graph.py:61-82 builds for its image counts); thr[G,3] = cutoff |recip_a| + margin;
 * reps[3] = images per axis of the enumeration (n_cells = prod (2 reps + 1), cartesian_prod order). */
int xeq_radius_graph_pbc_count_pruned(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                                      int64_t n_nodes, const void* img, int64_t n_cells, double cutoff, const void* recip,
                                      const void* thr, const int32_t reps[3], int32_t* deg, void* stream);
int xeq_radius_graph_pbc_fill_pruned(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs,
                                     int64_t n_nodes, const void* img, const void* cells, const void* shift, int64_t n_cells,
                                     double cutoff, const void* recip, const void* thr, const int32_t reps[3],
                                     const int32_t* rowptr, int64_t n_edges, int64_t* edge_index, void* cell_offsets,
                                     void* stream);

/* Cell-list form of the same search for graphs of many atoms (the pair sweep above is O(n_g^2)).  Bins live in
 * fractional coordinates: nbins[G,3] per axis (1 on open axes; bin width >= the cutoff across lattice planes, i.e.
 * nbins_a <= 1 / thr_a), bin_base[G+1] = running sum of bins per graph.  xeq_radius_graph_pbc_bin_ids writes each atom's
 * global bin id; the caller sorts atoms by it (xeq_csr_by_key: bin_start[n_bins+1], bin_atom[N]).  count/fill then
 * visit the 3x3x3 bin block around each center (same per-pair arithmetic and image pruning as the _pruned form); fill
 * writes unordered keys into tmp_keys[E] and ranks them per center, so edge_index / cell_offsets come out in the
 * reference's order, bit-identical to the exhaustive form. */
int xeq_radius_graph_pbc_bin_ids(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                                 const void* recip, const int32_t* nbins, const int32_t* bin_base, int64_t* keys, void* stream);
int xeq_radius_graph_pbc_count_cl(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                                  const void* img, int64_t n_cells, double cutoff, const void* recip, const void* thr,
                                  const int32_t reps[3], const int32_t* nbins, const int32_t* bin_base,
                                  const int32_t* bin_start, const int32_t* bin_atom, int32_t* deg, void* stream);
int xeq_radius_graph_pbc_fill_cl(int dtype, const void* pos_wrap, const int64_t* ptr, int64_t n_graphs, int64_t n_nodes,
                                 const void* img, const void* cells, const void* shift, int64_t n_cells, double cutoff,
                                 const void* recip, const void* thr, const int32_t reps[3], const int32_t* nbins,
                                 const int32_t* bin_base, const int32_t* bin_start, const int32_t* bin_atom,
                                 const int32_t* rowptr, int64_t n_edges, int64_t* tmp_keys, int64_t* edge_index,
                                 void* cell_offsets, void* stream);

/* ------------------------------------------------------------ edge geometry */

/* Neighbour-sorted permutation of a SYMMETRIC, center-sorted edge list whose neighbours are ascending and unique
 * within a center (the open-boundary builders above emit exactly that: replaces the stable sort by neighbour that
 * ops.EdgeGraph of the reference-side caller would otherwise need, cf. torch_scatter's index_add on edge_index[1],
 * nn/xpainn.py:154-159).  rev[e] = position of the reverse edge; n_perm = rev and n_rowptr = c_rowptr.
 * rev[e] = -1 where the reverse edge is missing (the list was not symmetric). */
int xeq_reverse_edge_map(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes, const int32_t* c_rowptr,
                         int32_t* rev, void* stream);
/* The same for a PERIODIC list of the builders below (xeq_radius_graph_pbc_*: center-sorted, a center's edges ascending in (neighbor,
 * image)): rev[e] = position of (j <- i, -offset) for e = (i <- j, offset), -1 where the list holds no such edge.  Such a list is
 * symmetric up to a rounding at the cutoff -- the reference forms |pos_i - (pos_j + image)| (data/radius_graph.py:117-121), and the two
 * directions round differently -- so a -1 can occur for an edge within an ulp of the cutoff, where the envelope is ~1e-14 (and its
 * slope ~1e-7 of the scale): the consumers of a mirror map (XEQ_WQ_MIRROR_WALK, xeq_message_wq_edge_grad*, xeq_edge_vectors_bwd) skip
 * such an entry, the edge then misses its vanishing reverse contribution.  n_edges may be a capacity (the list ends at c_rowptr[N]).
 * With it a periodic system's reverse pass walks the forward plan like an open one's: no sort by neighbor, no second plan, no second
 * set of records (round 5; ~0.1 ms of a 1 536-atom step). */
int xeq_reverse_edge_map_pbc(int dtype, const int64_t* edge_index, const void* cell_offsets, int64_t n_edges, int64_t n_nodes,
                             const int32_t* c_rowptr, int32_t* rev, void* stream);

/* compute_edge_data (nn/basic.py:110-131): vec = pos[c] - pos[n] - cell_offsets @ cell[batch[n]],
 * dist = |vec|.  cell/cell_offsets/batch may be NULL (non-PBC).  batch == NULL with a
 * cell means single graph (cell[0], nn/basic.py:121-123). */
int xeq_edge_vectors_fwd(int dtype, const void* pos, const int64_t* edge_index, int64_t n_edges,
                         const void* cell, const void* cell_offsets, const int64_t* batch, void* vec, void* dist,
                         void* stream);

/* Backward of the above w.r.t. pos, as a deterministic segmented sum (no atomics):
 * grad_pos[i] = sum_{e: center=i} gvec[e] - sum_{e: neighbor=i} gvec[e].
 * c_rowptr/c_perm: CSR over centers (perm NULL = edges already center-sorted);
 * n_rowptr/n_perm: CSR over neighbors. */
int xeq_edge_vectors_bwd(int dtype, const void* grad_vec, int64_t n_nodes, const int32_t* c_rowptr,
                         const int32_t* c_perm, const int32_t* n_rowptr, const int32_t* n_perm, void* grad_pos,
                         void* stream);

/* --------------------------------------------- operator-level (e3nn) drop-ins */

/* e3nn.o3.SphericalHarmonics(irreps, normalize=True, "component") as built at
 * nn/xpainn.py:49-51.  `vec` is in e3nn axis order (the caller already applied
 * vec[:, [1,2,0]], nn/xpainn.py:71-74).  out[E, D], every Y_l repeated mul[l] times. */
int xeq_sph_harm_fwd(int dtype, const void* vec, int64_t n, const int32_t mul[3], int normalize, void* out,
                     void* stream);
int xeq_sph_harm_bwd(int dtype, const void* vec, const void* grad_out, int64_t n, const int32_t mul[3],
                     int normalize, void* grad_vec, void* stream);

/* SphericalBesselj0 / GaussianSmearing (nn/rbf.py:114-152) and the envelopes
 * (nn/rbf.py:43-73) on dist[E].  rbf_out[E,B], fcut_out[E] (either may be NULL). */
int xeq_radial_fwd(int dtype, const void* dist, int64_t n, int rbf_kind, int cutoff_kind, int num_basis,
                   double cutoff, const void* p0, const void* p1, void* rbf_out, void* fcut_out, void* stream);

/* e3nn.o3.ElementwiseTensorProduct(irreps, "Cx0e") (nn/xpainn.py:119-121): out[n,u,m] = x[n,u,m]*g[n,u].
 * g_rows == 1 broadcasts one gate row (EquivariantLayerNorm affine, nn/o3layer.py:164). */
int xeq_elementwise_tp_fwd(int dtype, const void* x, const void* g, int64_t n, int64_t g_rows,
                           const int32_t mul[3], void* out, void* stream);

/* TensorProduct 'uuu' l x l -> 0e with path weight ir.dim (Invariant / EquivariantDot,
 * nn/o3layer.py:23-29,89-95): out[n,u] = sum_m a[n,u,m] b[n,u,m]. */
int xeq_channel_dot_fwd(int dtype, const void* a, const void* b, int64_t n, const int32_t mul[3], void* out,
                        void* stream);
/* grad_a[n,u,m] = g[n,u] * b[n,u,m] (same kernel as the elementwise TP). */

/* EquivariantLayerNorm.forward (nn/o3layer.py:145-171) and its backward w.r.t. x. */
int xeq_eqln_fwd(int dtype, const void* x, const void* weight, const void* bias, int64_t n,
                 const int32_t mul[3], double eps, void* out, void* stream);
int xeq_eqln_bwd(int dtype, const void* x, const void* weight, const void* grad_out, int64_t n,
                 const int32_t mul[3], double eps, void* grad_x, void* stream);

/* torch_scatter.scatter_sum over a sorted index given as ptr (nn/output.py:124):
 * out[g, :] = sum_{i in [ptr[g], ptr[g+1])} src[i, :]. */
int xeq_segment_sum(int dtype, const void* src, const int64_t* ptr, int64_t n_segments, int64_t width,
                    void* out, void* stream);
/* torch_scatter.scatter(src, index, dim=0, reduce="sum") for an arbitrary index
 * (float atomics; `out` must be zero-initialised by the caller). */
int xeq_scatter_add(int dtype, const void* src, const int64_t* index, int64_t n, int64_t width, void* out,
                    int64_t n_out, void* stream);

/* The model's first message block behind XEmbedding in ONE gather (round 5): there a node's scalar features, the LayerNorm /
 * scalar_mlp output h and the 0e block of the equivariant norm are functions of the node's ELEMENT alone (nn/xpainn.py:62, 76-81,
 * 128-139: s = Linear(table[Z]), x = 0), so they are evaluated once per table row (rows_s [Zmax + 1, node_dim], rows_h [., hidden_dim],
 * rows_x0 [., node_dim]) and gathered by atomic number: s_out [n, node_dim], h_out [n, hidden_dim], xhat_out [n * irreps_dim] in BT
 * layout with every l > 0 block zero.  z: int32 or int64 atomic numbers; one outside [0, n_rows) reads row 0.  Replaces three ATen
 * gathers and a fill. */
int xeq_first_block_front(const void* z, int z_is_int64, int64_t n, int64_t n_rows, const void* rows_s, const void* rows_h, const void* rows_x0,
                          int node_dim, int hidden_dim, int64_t irreps_dim, void* s_out, void* h_out, void* xhat_out, void* stream);

/* Up to XEQ_COPY_MANY_MAX device-to-device copies in ONE launch: dst[i][0 .. bytes[i]) = src[i][...] (whole aligned 4-byte
 * words, buffers must not overlap).  src / dst / bytes are HOST arrays.  HIP-graph replay (runtime.GraphedModel) refreshes
 * the captured inputs of an evaluation with it -- the per-step tensor hand-over the reference does with `data.to(device)`
 * (run/inference.py:39). */
/* Single f32 linear layers on the matrix cores, y = act(x W^T + b) (csrc/xeq_linear.hip): the node-side contractions that are
 * not two-layer MLPs -- dot_lin (nn/xpainn.py:191-193, :222-223) and its input gradient, the embedding Linear(56, 128) on the
 * gathered table rows (nn/xpainn.py:43-48, nn/basic.py:57: `row_index` = atomic numbers gathers the rows of x), the energy
 * head's first layer (nn/output.py:104-118).  w_packed: xeq_mlp_pack(W, b, n_out, k_in, transposed) -- transposed = 1 with
 * W given as [k_in][n_out] turns the same weight into the input-gradient product.  k_in % 8 == 0, <= 256; n_out % 32 == 0, <= 256;
 * act 0 none / 1 SiLU; pre (optional) receives the pre-activation.  A row's sums run in one fixed order whatever n is.
 * xeq_head_dot: out[n] = <hidden[n, :], w2> + b2 (the head's last layer, nn/output.py:104-106); xeq_head_bwd_hidden: its
 * reverse with the SiLU in front, g_hidden[n, j] = g_atomic[n] w2[j] silu'(pre[n, j]) (g_atomic NULL: ones). */
/* Test entry for the property the few-row forms rest on (csrc/xeq_linear_s.h): one 32 x 32 block of a [32, 224] x b [224, 32] through
 * v_mfma_f32_32x32x2_f32, through v_mfma_f32_16x16x4_f32 and as a sequential fmaf chain per element; diff[0] / diff[1] (int32, zeroed by the
 * caller) += the outputs whose bits differ between the two instruction shapes / between the matrix instruction and the chain. */
int xeq_mfma_order_probe(const float* a, const float* b, int32_t* diff, void* stream);
int xeq_linear_supported(int dtype, int k_in, int n_out);
int xeq_linear_fwd(const void* x, int64_t ldx, int64_t n, int k_in, const int32_t* row_index, const void* w_packed, int n_out,
                   int has_bias, int act, void* pre, void* y, int64_t ldy, void* stream);
int xeq_head_dot(const void* hidden, int64_t n, int hidden_dim, const void* w2, const void* b2, void* out, void* stream);
int xeq_head_bwd_hidden(const void* pre, int64_t n, int hidden_dim, const void* w2, const void* g_atomic, void* g_hidden, void* stream);
/* The whole energy head of a force evaluation in ONE launch (round 5; nn/output.py:104-128 with the reverse pass of
 * nn/basic.py:143-159): atomic[n] = <SiLU(W1 s_n + b1), w2> + b2 and, when `jac` is given, the head's reverse pass as a saved row
 * jac[n, :] = d atomic_n / d s_n = W1^T (w2 . SiLU'(W1 s_n + b1)) -- the output is one number per node, so the reverse pass of
 * whatever reaches the head is g_s[n, :] = (g_atomic[n] + g_total[batch[n]]) jac[n, :] (xeq_head_bwd; either gradient may be NULL,
 * strides in elements, 0 = a broadcast scalar; batch NULL: one graph).  w1_packed = xeq_mlp_pack(W1, b1, hidden, node_dim, 0),
 * w1t_packed = xeq_mlp_pack(W1, NULL, node_dim, hidden, 1).  Widths: multiples of 32, <= 256.  Replaces xeq_linear_fwd + xeq_head_dot
 * (+ xeq_segment_sum's reverse gather) + xeq_head_bwd_hidden + xeq_linear_fwd of the round-3 head: seven launches -> three. */
int xeq_head_supported(int dtype, int node_dim, int hidden_dim);   /* 1 / 0, not a status */
int xeq_head_fwd(const void* s, int64_t lds, int64_t n, int node_dim, int hidden_dim, const void* w1_packed, const void* w1t_packed,
                 const void* w2, const void* b2, void* atomic, void* jac, void* stream);
int xeq_head_bwd(const void* jac, int64_t n, int node_dim, const void* g_atomic, int64_t ga_stride, const void* g_total, int64_t gt_stride,
                 const int64_t* batch, void* g_s, void* stream);

/* Weight gradient of a linear layer for a TRAINING pass (what autograd forms as grad_output^T @ input for nn.Linear, utils/trainer.py:
 * 295-302; the o3.Linear blocks likewise): parts[c][M][K] = sum over the rows of chunk c of a[row, 0..M)^T b[row, 0..K), f32, rows
 * with strides lda / ldb; n_chunks = xeq_wgrad_chunks(n, M, K); dW = sum over c (fixed order: reproducible bit for bit).  with_bias:
 * every chunk's block is followed by M more floats, the column sums of a over the chunk's rows (the layer's bias gradient): parts is
 * [n_chunks][M K + M]. */
int xeq_wgrad_chunks(int64_t n, int m, int k);
int xeq_wgrad(const void* a, int64_t lda, const void* b, int64_t ldb, int64_t n, int m, int k, int with_bias, int n_chunks, void* parts,
              void* stream);

/* A batch of n atoms in g graphs into arrays of n_cap atoms / g_cap graphs in ONE launch (runtime.GraphedStep: neighbour list +
 * model as one captured graph over capacity-sized arrays): atoms n .. n_cap - 1 get atomic number 0, positions
 * (pad0 + spacing (i - n), 0, 0) -- no two within any cutoff -- and sit alone in graph g_cap - 1; graphs g .. g_cap - 2 are empty.
 * ptr [g + 1] / ptr_out [g_cap + 1], batch [n] / batch_out [n_cap] int64; batch may be NULL: an atom's graph is then found in ptr. */
int xeq_load_padded_batch(int dtype, const void* pos, const int32_t* z, const int64_t* ptr, const int64_t* batch, int64_t n,
                          int64_t g, int64_t n_cap, int64_t g_cap, double pad0, double spacing, void* pos_out, int32_t* z_out,
                          int64_t* ptr_out, int64_t* batch_out, void* stream);
/* the same reading int64 atomic numbers (a torch.long tensor as it is: no conversion launch in front of every step) */
int xeq_load_padded_batch_z64(int dtype, const void* pos, const int64_t* z, const int64_t* ptr, const int64_t* batch, int64_t n, int64_t g,
                              int64_t n_cap, int64_t g_cap, double pad0, double spacing, void* pos_out, int32_t* z_out,
                              int64_t* ptr_out, int64_t* batch_out, void* stream);
/* a contiguous range of molecules of a larger batch (a shard of it: runtime.GraphedLanes): pos, z, batch point at the range's first
 * atom, ptr at its first graph; the range is renumbered from zero (ptr values - atom_offset, batch values - graph_offset) */
int xeq_load_padded_shard(int dtype, const void* pos, const void* z, int z_is_int64, const int64_t* ptr, const int64_t* batch, int64_t n,
                          int64_t g, int64_t atom_offset, int64_t graph_offset, int64_t n_cap, int64_t g_cap, double pad0, double spacing,
                          void* pos_out, int32_t* z_out, int64_t* ptr_out, int64_t* batch_out, void* stream);

#define XEQ_COPY_MANY_MAX 16
int xeq_copy_many(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream);
/* The same number of buffer pairs COMPARED in one launch (runtime.GraphedModel: is the engine's neighbour list the one the captured
 * graph holds?): flag[0] = gen when any 4-byte word of any pair differs, untouched otherwise; gen: a value the flag has never held. */
int xeq_compare_many(int n, const void* const* a, const void* const* b, const int64_t* bytes, int32_t gen, int32_t* flag, void* stream);

/* ALL instructions of a general Clebsch-Gordan e3nn.o3.TensorProduct in one launch (round 4; nn/tp.py:20-107 builds the instruction lists;
 * nn/xe3net.py:133-146 'uuu' with shared weights, nn/output.py:411-421 'uuw' with one weight set per sample):
 *   out[n, off_out + w (2 l3 + 1) + k] = sum_paths coeff_p sum_{u,v} W_p[..] sum_{i,j} cg_p[i,j,k] x1[n, off1 + u (2 l1 + 1) + i] x2[n, off2 + v (2 l2 + 1) + j]
 * A workgroup stages the x1 / x2 rows of a tile of nodes and every 3j table in LDS; a thread owns an output element (a block of four w for
 * 'uvw' / 'uuw', where the bilinear form is shared by all w) and walks the paths into its block in table order: one store per element, no
 * read-modify-write, deterministic.  paths: n_paths x 10 ints (off1, off2, off_out, mul1, mul2, mul_out, l1, l2, l3, mode: 0 uvw, 1 uvu,
 * 2 uvv, 3 uuw, 4 uuu, 5 uvuv), sorted by off_out, at most 40 paths into at most 16 blocks per launch; cg: the paths' real Wigner-3j tables
 * one behind the other (device, dtype of x), cg_off[p] / cg_floats; w_off[p]: first weight of path p in `weight` (-1: unweighted), shared
 * (weight_stride 0) or per sample; coeff[p] (host).  The reverse passes w.r.t. x1 / x2 are the same entry on (grad_out, x2) / (x1, grad_out)
 * with permuted tables and the connection mode of the transposed contraction (xequinet_amd/tp.py::_REVERSE).
 * xeq_tensor_product_wgrad: dL/dW of the weighted paths (same table; w_off increasing, w_numel[p] weights per path): shared = 1 writes
 * parts [xeq_tensor_product_wgrad_chunks(n), w_total] (the caller adds the chunks in order), shared = 0 writes dW [n, w_total]. */
int xeq_tensor_product(int dtype, const void* x1, const void* x2, int64_t n, int dim1, int dim2, int dim_out, int n_paths,
                       const int32_t* paths, const void* cg, const int32_t* cg_off, int cg_floats, const void* weight, int64_t weight_stride,
                       const int32_t* w_off, const double* coeff, void* out, void* stream);
int64_t xeq_tensor_product_wgrad_chunks(int64_t n);
int xeq_tensor_product_wgrad(int dtype, const void* x1, const void* x2, const void* g, int64_t n, int dim1, int dim2, int dim_out, int n_paths,
                             const int32_t* paths, const void* cg, const int32_t* cg_off, const int32_t* w_off, const int32_t* w_numel, int w_total,
                             const double* coeff, int shared, void* parts, void* stream);

/* One instruction of a general Clebsch-Gordan e3nn.o3.TensorProduct (nn/tp.py:20-107 builds the instruction lists;
 * nn/xe3net.py:133-146 'uuu', nn/output.py:411-421 'uuw'):
 *   out[n, off_out + w (2 l3 + 1) + k] += coeff * sum_{u,v} W[..] sum_{i,j} cg[i,j,k] x1[n, off1 + u (2 l1 + 1) + i] x2[n, off2 + v (2 l2 + 1) + j]
 * mode: 0 uvw, 1 uvu, 2 uvv, 3 uuw, 4 uuu, 5 uvuv (e3nn connection modes: which of u, v, w are tied); cg = the real
 * Wigner-3j table [2 l1 + 1, 2 l2 + 1, 2 l3 + 1] (device memory, dtype of x); weight NULL = unweighted path, otherwise
 * the path's weights in e3nn's order, shared (weight_stride 0) or one set per sample (weight_stride = floats per sample).
 * The launches of a product's paths accumulate into a zero-initialised `out` in stream order: deterministic. */
int xeq_tensor_product_path(int dtype, const void* x1, const void* x2, int64_t n, int dim1, int dim2, int dim_out, int off1,
                            int off2, int off_out, int mul1, int mul2, int mul_out, int l1, int l2, int l3, int mode,
                            const void* cg, const void* weight, int64_t weight_stride, double coeff, void* out, void* stream);

/* ------------------------------------------------------------ fused message */

/* GENERIC form (64-bit offsets, f32 / f64, up to 256 channels per kind): the fallback for sizes and layouts the
 * _wq and _sb forms below do not take.
 * XPainnMessage.forward lines nn/xpainn.py:140-159 in one pass over
 * destination-sorted edges (K5-K7, K11-K16 of SURVEY 2.2):
 *   filter = (rbf(d) W^T + b) * fcut(d)                       :140
 *   g = h[nbr] * filter = [gate_state C | gate_edge C | msg_s F]  :142-148
 *   msg_x = xhat[nbr] (x) gate_state + Y(vec) (x) gate_edge    :150-154
 *   s_out = s_in + sum_{e->c} msg_s ; x_out = x_in + sum msg_x  :158-159
 * rbf / fcut / Y_lm are recomputed from vec per edge and never materialised.
 * h[N, 2C+F] = scalar_mlp output (:139), xhat[N, D] = o3norm output (:131).
 * rowptr[N+1]/perm[E]: CSR over centers (perm NULL = already sorted);
 * nbr = edge_index row 1 (int64[E]); vec[E,3] in ORIGINAL axis order.
 * Per-node segmented sum in registers: no atomics, bitwise reproducible. */
int xeq_message_fwd(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                    const int64_t* nbr, const void* vec, const void* h, const void* xhat, const void* s_in,
                    const void* x_in, const void* w_rbf, const void* b_rbf, const void* p0, const void* p1,
                    int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                    const int32_t mul[3], void* s_out, void* x_out, int xhat_layout, void* stream);

/* Reverse pass of the fused message w.r.t. h, xhat and vec (what the force
 * evaluation nn/basic.py:143-159 needs; parameter gradients are out of scope).
 * Iterates the CSR over NEIGHBORS (n_rowptr/n_perm; n_perm NULL = edges sorted by
 * neighbor) so grad_h / grad_xhat are segmented sums too; grad_vec[E,3] is written
 * at the edge's own position.  center = edge_index row 0. */
int xeq_message_bwd(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                    const int64_t* center, const void* vec, const void* h, const void* xhat, const void* grad_s,
                    const void* grad_x, const void* w_rbf, const void* b_rbf, const void* p0, const void* p1,
                    int rbf_kind, int cutoff_kind, int num_basis, double cutoff, int node_dim,
                    const int32_t mul[3], void* grad_h, void* grad_xhat, void* grad_vec, int xhat_layout,
                    void* stream);

/* Parameter gradients of the radial filter for a TRAINING pass (rbf_lin.weight / .bias, nn/xpainn.py:117,140; the trainable
 * basis parameters freq, nn/rbf.py:143-146, or mean / std, :121-125) -- what autograd gets in the reference by differentiating
 * through the materialised filter[E, 576] (nn/basic.py:154-155 with create_graph=training, utils/trainer.py:295-302).  Same walk
 * and operands as xeq_message_bwd (neighbor-sorted CSR, center = edge_index row 0, grad_s / grad_x = dL/ds_out, dL/dx_out);
 * writes parts[n_parts][H][3 B + 1] (n_parts = xeq_message_param_grad_parts(n_nodes)), per filter row c:
 *   [0, B)        sum_e G f rho_k             -> dL/dW[c, k] after the sum over parts
 *   [B]           sum_e G f                   -> dL/db[c]
 *   [B+1, 2B+1)   sum_e G f d rho_k / d p0_k  -> dL/dp0_k = sum_c W[c, k] (.)
 *   [2B+1, 3B+1)  sum_e G f d rho_k / d p1_k  -> dL/dp1_k likewise (zeros for the Bessel basis)
 * with G[e, c] = dL/dfilter_out[e, c] h[nbr(e), c].  f32 and f64; node_dim and the channel count at most 256 each. */
int xeq_message_param_grad_parts(int64_t n_nodes);
int xeq_message_param_grad(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                           const int64_t* center, const void* vec, const void* h, const void* xhat, const void* grad_s,
                           const void* grad_x, const void* p0, const void* p1, int rbf_kind, int cutoff_kind, int num_basis,
                           double cutoff, int node_dim, const int32_t mul[3], int xhat_layout, int n_parts, void* parts,
                           void* stream);

/* The same parameter gradients on the matrix cores (csrc/xeq_train.hip; f32, node_dim and multiplicities in multiples of 32, a table
 * row of 64 floats: xeq_message_param_grad_mc_supported).  xeq_param_basis writes, once per training step, the per-edge rows
 *   tab[e][W] = f(d_e) [rho_k (B) | 1 | d rho_k/d p0 (B) | d rho_k/d p1 (B, gaussian basis only)], zero padding to a multiple of 4, Y_1 (3), Y_2 (5)
 * (W = xeq_param_basis_width: whole 128-byte lines) in the order of the edge list; xeq_message_param_grad_mc contracts them with G[e, c] formed on the fly:
 * parts[n_parts][H][64], columns as xeq_message_param_grad's (the caller sums over the parts).  center / nbr = edge_index rows 0 / 1;
 * any edge order (center-sorted lists gather best).  n_valid (optional, device pointer): the edge count of a capacity-sized list whose
 * size never reached the host (a training step captured as one HIP graph, train.GraphedTrainStep): edges behind it are not walked. */
int xeq_message_param_grad_mc_supported(int dtype, int rbf_kind, int num_basis, int node_dim, const int32_t mul[3]);
int xeq_param_basis_width(int rbf_kind, int num_basis);
int xeq_message_param_grad_mc_parts(int64_t n_edges, int node_dim, const int32_t mul[3]);
int xeq_param_basis(const void* vec, int64_t n_edges, int rbf_kind, int cutoff_kind, int num_basis, double cutoff, const void* p0,
                    const void* p1, void* tab, void* stream);
int xeq_message_param_grad_mc(int64_t n_nodes, int64_t n_edges, const int64_t* center, const int64_t* nbr, const void* tab, const void* h,
                              const void* xhat, const void* grad_s, const void* grad_x, int rbf_kind, int num_basis, int node_dim,
                              const int32_t mul[3], int xhat_layout, int grad_x_layout, const int32_t* n_valid, int n_parts, void* parts,
                              void* stream);
/* Affine-parameter gradients of a block's two norms for a TRAINING pass (nn.LayerNorm weight / bias on the scalars,
 * EquivariantLayerNorm affine_weight / affine_bias, nn/o3layer.py:145-171), f32: from the block inputs s [n, F], x [n, D], the
 * statistics stats [n, 4] the forward norm kernel wrote, dL/dshat rows (stride ld_gs) and dL/dxhat in the BT layout.
 * parts[n_chunks][2 F + C + mul0] = [d ln_w | d ln_b | d eq_w | d eq_b] per row chunk (n_chunks = xeq_norm_param_grad_chunks(n));
 * the caller sums the chunks (fixed order). */
int xeq_norm_param_grad_chunks(int64_t n_nodes);
int xeq_norm_param_grad(const void* s, const void* x, const void* stats, const void* g_shat, int64_t ld_gs, const void* g_xhat_bt,
                        int64_t n_nodes, int node_dim, const int32_t mul[3], int n_chunks, void* parts, void* stream);
/* f32 x[n_nodes, D] in the e3nn mul_ir layout -> the internal BT layout (per l a row-major [N (2l+1), mul_l] matrix; layout 1 of the
 * xhat_layout / grad_x_layout arguments): the gradient kernel's gathers of dL/dx_out rows are contiguous over the channels there. */
int xeq_to_bt(const void* x, int64_t n_nodes, const int32_t mul[3], void* out, void* stream);

/* Node-side functions of a training pass that is differentiated TWICE (forces in the loss: nn/basic.py:143-159 with create_graph=training,
 * utils/trainer.py:295-308; host side ops.NormFn / UvFn / UpdateOutFn, csrc/xeq_train_node.hip).  f32 / f64.  Every entry has a forward
 * form (reverse = 0) and a reverse form (reverse = 1); with the *_tan pointers given, the same body runs on dual numbers (value, tangent)
 * and the outputs are the TANGENTS -- that is the second order: for the cotangent u of a first-order input gradient, the forward form at
 * tangent u gives d<u, xbar>/dg, the reverse form at tangent u gives d<u, xbar>/dx and d<u, xbar>/dtheta.
 *
 * xeq_train_norm: (s[N, F], x[N, D]) -> (LayerNorm(s), EquivariantLayerNorm(x))  (nn/xpainn.py:130-131, :213-214; nn/o3layer.py:145-171:
 *   the l = 0 block is centred, one rms over all C channels, weight per channel, bias on the l = 0 block).  xhat_layout 0: the e3nn row,
 *   1: BT (per l a row-major [N (2l+1), mul_l] matrix, one after the other) -- of out_x (forward) and of g_x (reverse).
 *   reverse: out_s / out_x = dL/ds, dL/dx (e3nn rows); rows[N, 2F + C + mul0] = per-node [d ln_w | d ln_b | d eq_w | d eq_b] (the caller sums).
 * xeq_train_uv: uv[l] = [N (2l+1), 2 mul_l] row-major (U | V of update_U / update_V in BT rows) -> out[N, 2C] = [Invariant(V) | sum_m U V]
 *   (nn/o3layer.py:40-44, :61-68; eps of the Invariant).  reverse: g = dL/dout, d_uv[l] = dL/duv[l].
 * xeq_train_out: (uv, a[N, C + 2F] = [a_vv | a_sv | a_ss], inner[N, F]) -> out0[N, F] = a_sv inner + a_ss, out1[N, D] = U a_vv (e3nn row)
 *   (nn/xpainn.py:226-230 without the residual).  reverse: g_s, g_x = dL/dout0, dL/dout1; out0 = dL/da, out1 = dL/dinner, d_uv. */
int xeq_train_norm(int dtype, int reverse, int64_t n, const void* s, const void* s_tan, const void* x, const void* x_tan, const void* ln_w,
                   const void* ln_b, const void* eq_w, const void* eq_b, const void* g_s, const void* g_x, int node_dim,
                   const int32_t mul[3], double eps_ln, double eps_eq, int xhat_layout, void* out_s, void* out_x, void* rows, void* stream);
int xeq_train_uv(int dtype, int reverse, int64_t n, const void* const uv[3], const void* const uv_tan[3], const void* g, const int32_t mul[3],
                 double eps, void* out, void* const d_uv[3], void* stream);
int xeq_train_out(int dtype, int reverse, int64_t n, const void* const uv[3], const void* const uv_tan[3], const void* a, const void* a_tan,
                  const void* inner, const void* inner_tan, const void* g_s, const void* g_x, int node_dim, const int32_t mul[3],
                  void* out0, void* out1, void* const d_uv[3], void* stream);

/* Launch policy, stated ONCE for every front (the Python modules' ops.select_message_impl / ops._wq_edges_per_stream and the
 * registered operator xeq::xpainn_eval call these): the kernel family `auto` takes for a configuration and these sizes, and the
 * stream length of a wq walk plan (n_ranges = ceil(E / (2 x edges_per_stream))). */
#define XEQ_FAMILY_WQ 0
#define XEQ_FAMILY_SB 1
#define XEQ_FAMILY_GENERIC 3
int xeq_message_auto_family(int dtype, int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]);
int xeq_message_wq_edges_per_stream(int64_t n_nodes, int64_t n_edges);

/* "Scalar broadcast" form of the fused message (default path, f32 and f64).  The per-edge quantities
 * every channel shares -- f*rho_k(d), f, Y_lm, and their d/dd companions -- are evaluated once per
 * model evaluation into 4*(roundup(B,4)+12)-byte records (xeq_edge_basis), shared by all message
 * blocks and both directions; the message kernels fetch them with scalar loads (the edge index is
 * workgroup-uniform) and need no LDS and no barriers in the forward pass.
 * basis[E, W] / dbasis[E, W], W = xeq_edge_basis_width(B); dbasis may be NULL when only the forward
 * pass is needed.  Same semantics as xeq_message_fwd / xeq_message_bwd otherwise; s_in / x_in of xeq_message_fwd_sb may be NULL
 * (the aggregate without the residual: what the training pass's ops.DiffMessage asks for). */
int xeq_edge_basis_width(int num_basis);
/* 1 / 0: at most 256 channels per kind, num_basis <= 32, 32-bit element offsets (n_nodes max(H, D) < 2^31) */
int xeq_message_sb_fits(int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]);
int xeq_edge_basis(int dtype, const void* vec, int64_t n_edges, int rbf_kind, int cutoff_kind, int num_basis,
                   double cutoff, const void* p0, const void* p1, void* basis, void* dbasis, void* stream);
int xeq_message_fwd_sb(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm,
                       const int64_t* nbr, const void* basis, const void* h, const void* xhat, const void* s_in,
                       const void* x_in, const void* w_rbf, const void* b_rbf, int num_basis, int node_dim,
                       const int32_t mul[3], void* s_out, void* x_out, int xhat_layout, void* stream);
int xeq_message_bwd_sb(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                       const int64_t* center, const void* basis, const void* dbasis, const void* h, const void* xhat,
                       const void* grad_s, const void* grad_x, const void* w_rbf, const void* b_rbf, int num_basis,
                       int node_dim, const int32_t mul[3], void* grad_h, void* grad_xhat, void* grad_vec,
                       int xhat_layout, void* stream);

/* The reverse pass in the form a TRAINING pass differentiates a second time (force loss: nn/basic.py:143-159 with create_graph=training,
 * utils/trainer.py:295-308; host side ops.DiffMessage).  The message nn/xpainn.py:140-159 is multilinear in (dL/dout, h, xhat | Y, the
 * record head f rho_k | f, the rbf_lin rows), so each second-order term is xeq_message_fwd_sb or this entry with ONE operand replaced by
 * its tangent.  Records as xeq_edge_basis writes them (here formed by the caller, differentiably).  Walk and operands of xeq_message_bwd_sb;
 * instead of dL/dvec it writes
 *   q[E, 2C+F]  = h[nbr(e), c] P[e, c]   (P = <dL/dx_out[center], xhat[nbr]>, <dL/dx_out[center], Y_e>, dL/ds_out[center]) -- the caller gets
 *                 dL/d(record head) = q [W | b] and dL/d[W | b] = q^T (record head) from two library GEMMs;
 *   gy[E, 8]    = dL/dY_1[3], dL/dY_2[5] of the edge.
 * flags: bit 0 the xhat layout, XEQ_SB_Y0_ZERO (also accepted by xeq_message_fwd_sb's xhat_layout): the l = 0 harmonic counts as 0 -- the
 * record holds a tangent of the harmonics; XEQ_SB_Q_ACCUMULATE: q += instead of q =; XEQ_SB_NO_GY.  f32 / f64. */
#define XEQ_SB_Y0_ZERO 8
/* xeq_message_bwd_sb only: xhat_layout | XEQ_SB_ACCUM_VEC ADDS dL/dvec to what grad_vec holds instead of storing it: the message blocks of
 * one force evaluation share one buffer (the last block of the model stores, the earlier ones add -- the order in which autograd would
 * sum their separate results, without its two elementwise launches per evaluation). */
#define XEQ_SB_ACCUM_VEC 16
#define XEQ_SB_Q_ACCUMULATE 16
#define XEQ_SB_NO_GY 32   /* gy is not wanted (and not written) */
int xeq_message_bwd_sbq(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                        const int64_t* center, const void* basis, const void* h, const void* xhat, const void* grad_s,
                        const void* grad_x, const void* w_rbf, const void* b_rbf, int num_basis, int node_dim,
                        const int32_t mul[3], void* grad_h, void* grad_xhat, void* q, void* gy, int flags, void* stream);
/* dL/d[W | b] from those products: parts[n_chunks][2C+F][roundup(B, 4) + 1] with out[c, k] = sum_e q[e, c] basis[e, k] over the chunk's
 * edges (columns [0, B): dL/dW[c, k]; column roundup(B, 4): dL/db[c]); the caller adds the chunks in order.  At most 768 filter rows. */
/* Two of the three second-order passes in one walk each: pass A (h <- u_h) and pass B ((xhat, Y) <- (u_xhat, u_Y); the l = 0 harmonic counts as
 * 0, the scalar message has no term) use the TRUE record head, i.e. the same filter values and the same gathered rows; the pair kernels
 * evaluate the filter once per edge.  basis_u: records whose harmonics are u_Y (only their tail is read).
 *   xeq_message_fwd_sb_pair:  s_out = s_in + A's scalar aggregate, x_out = x_in + A's + B's equivariant aggregates (s_in / x_in may be NULL);
 *   xeq_message_bwd_sbq_pair: grad_h = B's dL/dh, grad_xhat = A's dL/dxhat, q = q_A + q_B, gy = A's dL/dY. */
int xeq_message_fwd_sb_pair(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* perm, const int64_t* nbr,
                            const void* basis, const void* basis_u, const void* h, const void* u_h, const void* xhat, const void* u_xhat,
                            const void* s_in, const void* x_in, const void* w_rbf, const void* b_rbf, int num_basis, int node_dim,
                            const int32_t mul[3], void* s_out, void* x_out, int xhat_layout, void* stream);
int xeq_message_bwd_sbq_pair(int dtype, int64_t n_nodes, int64_t n_edges, const int32_t* n_rowptr, const int32_t* n_perm,
                             const int64_t* center, const void* basis, const void* basis_u, const void* h, const void* u_h, const void* xhat,
                             const void* u_xhat, const void* grad_s, const void* grad_x, const void* w_rbf, const void* b_rbf, int num_basis,
                             int node_dim, const int32_t mul[3], void* grad_h, void* grad_xhat, void* q, void* gy, int flags, void* stream);
/* Rows [*n_valid, n_rows) of a row-major buffer of 4-byte words := 0 (n_valid: device pointer).  For q and a capacity-sized edge list whose
 * true count never reached the host (train.GraphedTrainStep): xeq_message_bwd_sbq writes the rows of walked edges only, the two products
 * above read every row. */
int xeq_zero_rows_from(void* buf, int64_t row_words, int64_t n_rows, const int32_t* n_valid, void* stream);
int xeq_message_q_wgrad_chunks(int64_t n_edges);
int xeq_message_q_wgrad(int dtype, const void* q, const void* basis, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3],
                        int n_chunks, void* parts, void* stream);

/* "Wave / quad" form of the fused message (f32; the default whenever the channel layout allows it: node_dim == mul[0], mul[l] % 32 == 0,
 * num_basis <= 31).  The filter (rbf_lin, nn/xpainn.py:117,140: [2C+F, B+1] x [B+1] per edge) runs on the matrix cores with the channel
 * on the lane (the first sixteen basis functions as split-bf16 products, the rest and the bias column as exact-f32 steps); each half-wave
 * is a STREAM over a contiguous range of CSR segments, so the aggregation (index_add, nn/xpainn.py:158-159) is a running sum in registers
 * that is stored once per node: deterministic, no atomics, no read-modify-write (nn/xpainn.py:140-159 and its reverse pass), on a
 * PADDED walk order: every node's edge list is padded to a multiple of four slots (a "quad" = the four accumulator
 * registers 4g..4g+3 of a lane; padding slots carry an all-zero record), so a quad belongs to one node and the segment
 * logic (reset / residual / the node's only store) runs once per quad instead of once per row.
 *   xeq_message_wq_plan: the walk plan of one direction, from the CSR of the walk order (rowptr[N+1], perm[E] = edge id
 *     per slot or NULL), owner / gather = the edge_index rows the walk is sorted by / gathers from (forward: center /
 *     neighbor; reverse: neighbor / center).  Outputs, sized for P = xeq_message_wq_pcap(N, E) padded slots:
 *     qptr[N+1] (first quad per node), pgath[P] (gathered node per padded slot), peid[P] (edge id, -1 for pads),
 *     qinfo[P/4] (owner | first << 30 | last << 31 per quad), sq / sn[2 n_ranges + 1] (quad / node boundaries of the
 *     streams, on segment starts at about equal quad counts), win[xeq_message_wq_win_ints(n_ranges)] (per STEP = xeq_message_wq_waves()
 *     consecutive ranges, walked together by the waves of a workgroup: first gathered node and row count of the window that
 *     holds every node the step gathers from; since round 6 for two stream CLASSES -- the table's streams, walked by the l > 0
 *     units, and for large batches streams of three table streams each, walked by the l = 0 units -- one table behind the
 *     other).  workspace: xeq_message_wq_plan_workspace(N) bytes.
 *     Depends on the graph only, not on the positions.
 *   xeq_edge_basis_wq: per-edge records IN PADDED WALK ORDER, basis / dbasis [P, xeq_message_wq_record_floats_for(num_basis)] floats
 *     ([12 even k | 12 odd k | Y1[3] Y2[5]], value and d/dd; dbasis may be NULL), once per evaluation and direction.
 *   xeq_message_fwd_wq / _bwd_wq: one launch per message block and direction.  A workgroup walks a chunk of steps; per step it stages the window's
 *     rows (the unit's columns of h and xhat, or of grad_x and grad_s) in LDS with 16-byte loads when they fit 48 KB --
 *     batches of molecules: a few molecules -- and the per-row gathers become LDS reads; a step whose window does not fit
 *     (periodic systems, large graphs) gathers from global memory. rowptr is the CSR of the walk order (nodes without edges keep
 *     s_in / x_in, or get zero gradients).  parts[xeq_message_wq_parts_floats(N, E, mul)]: per-unit partials of dL/dd
 *     and dL/dY_lm by padded slot of the reverse walk; xeq_message_wq_edge_grad sums them in fixed unit order into
 *     grad_vec[E,3] at the edge's own position. */
int xeq_message_wq_supported(int num_basis, int node_dim, const int32_t mul[3]);   /* 1 / 0, not a status */
int xeq_message_wq_fits(int64_t n_nodes, int64_t n_edges, int num_basis, int node_dim, const int32_t mul[3]);   /* 1 / 0 */
int64_t xeq_message_wq_pcap(int64_t n_nodes, int64_t n_edges);                      /* a size, not a status */
int xeq_message_wq_waves(void);
int xeq_message_wq_record_floats_for(int num_basis);   /* floats per padded slot of a record buffer: 40 up to 23 basis functions, 48 up to 31 */
int xeq_message_wq_record_floats(void);   /* floats per padded slot of a record buffer (basis / dbasis of xeq_edge_basis_wq) */   /* ranges per step (= waves per workgroup): win holds 2 ceil(n_ranges / that) entries */
int64_t xeq_message_wq_plan_workspace(int64_t n_nodes);                             /* bytes, -1 on failure */
int64_t xeq_message_wq_win_ints(int n_ranges);                                      /* ints of a plan's window table `win` (a size, not a status) */
int xeq_message_wq_plan(const int32_t* rowptr, const int32_t* perm, const int64_t* owner, const int64_t* gather,
                        int64_t n_nodes, int64_t n_edges, int n_ranges, void* workspace, int64_t workspace_bytes,
                        int32_t* qptr, int32_t* pgath, int32_t* peid, int32_t* qinfo, int32_t* sq, int32_t* sn, int32_t* win,
                        void* stream);
int xeq_edge_basis_wq(const void* vec, int64_t n_nodes, int64_t n_edges, const int32_t* qptr, const int32_t* peid,
                      int rbf_kind, int cutoff_kind, int num_basis, double cutoff, const void* p0, const void* p1,
                      void* basis, void* dbasis, void* stream);
int xeq_message_fwd_wq(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* sq, const int32_t* sn, const int32_t* win,
                       const int32_t* c_rowptr, const int32_t* pgath, const int32_t* qinfo, const void* basis, const void* h,
                       const void* xhat, const void* s_in, const void* x_in, const void* w_rbf, const void* b_rbf,
                       int num_basis, int node_dim, const int32_t mul[3], void* s_out, void* x_out, int xhat_layout,
                       void* stream);
int64_t xeq_message_wq_parts_floats(int64_t n_nodes, int64_t n_edges, const int32_t mul[3]);   /* a size, not a status */
/* grad_h / grad_xhat may both be NULL: only the per-edge partials (dL/dvec) are formed.
 * xhat_layout of the _wq entries carries an optional hint: xhat_layout | XEQ_XHAT_HIGHER_L_ZERO promises that xhat is zero on
 * every l > 0 column.  That is the model's first message block (XEmbedding hands over x = 0, nn/xpainn.py:76-81, and the
 * equivariant layer norm of zero is zero off the 0e columns): the forward kernel then drops the gate_state term of the l > 0
 * units with its filter and gathers (bit-identical results), the reverse kernel (with grad_h NULL) drops the value filters
 * and pass S of the l > 0 units.  The other kernel families accept the hint and ignore it. */
#define XEQ_XHAT_HIGHER_L_ZERO 2
/* xeq_message_bwd_wq only: xhat_layout | XEQ_WQ_MIRROR_WALK walks the FORWARD plan (center-sorted CSR, owner = center, gathered =
 * neighbor) and its records: every slot (i <- j) then stands for its mirror edge (j <- i), whose neighbor is the owner and whose
 * center is the gathered node -- the edges a reverse walk visits for node i, in the same order, when the list is symmetric with
 * neighbours ascending per center (what the open-boundary builders of this library write).  The mirror's vector is the negative of
 * the slot's: same distance, same Y_2, Y_1 negated (done at the load), so the results are the reverse plan's bit for bit, and the
 * reverse plan and its records are never built.  xeq_message_wq_edge_grad then takes `mirror` = the reverse-edge map
 * (xeq_reverse_edge_map: position of edge (j, i) for edge (i, j)) to write each slot's gradient to the mirror edge; NULL otherwise. */
#define XEQ_WQ_MIRROR_WALK 4
/* xeq_message_fwd_wq / _bwd_wq: xhat_layout | XEQ_WQ_PACKED_WEIGHTS says that `w_rbf` points at the output of
 * xeq_message_wq_pack_weights (the units' rbf_lin rows -- bf16 splits of the first sixteen basis columns, the exact-f32 tail and the
 * bias -- in the kernels' LDS layout, xeq_message_wq_packed_weight_floats floats, once per weight version) and `b_rbf` is ignored: the
 * workgroups then stage their unit's weights with coalesced 16-byte loads instead of one row per lane (~11 us per launch). */
#define XEQ_WQ_PACKED_WEIGHTS 64
int64_t xeq_message_wq_packed_weight_floats(int num_basis, int node_dim, const int32_t mul[3]);   /* a size (-1: unsupported), not a status */
int xeq_message_wq_pack_weights(const void* w_rbf, const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* packed,
                                void* stream);
int xeq_message_bwd_wq(int64_t n_nodes, int64_t n_edges, int n_ranges, const int32_t* sq, const int32_t* sn, const int32_t* win,
                       const int32_t* n_rowptr, const int32_t* pgath, const int32_t* qinfo, const void* basis,
                       const void* dbasis, const void* h, const void* xhat, const void* grad_s, const void* grad_x,
                       const void* w_rbf, const void* b_rbf, int num_basis, int node_dim, const int32_t mul[3], void* grad_h,
                       void* grad_xhat, void* parts, int xhat_layout, void* stream);
int xeq_message_wq_edge_grad(const void* vec, int64_t n_nodes, int64_t n_edges, const int32_t* qptr, const int32_t* peid,
                             const int32_t* mirror, const int32_t mul[3], const void* parts, void* grad_vec, void* stream);
/* The same for the partials of SEVERAL message blocks of one evaluation (all walked on the same plan): parts[0 .. n_sets) are summed
 * per quantity in that order before the chain rule runs once (it is linear in dL/dd, dL/dY_lm): the three blocks' contributions to
 * dL/dvec -- which autograd otherwise adds up with elementwise launches behind three edge-gradient launches, nn/basic.py:143-159 over
 * nn/xpainn.py:140-159 -- in ONE launch.  n_sets <= XEQ_WQ_MAX_PART_SETS; `parts` is a HOST array of device pointers. */
#define XEQ_WQ_MAX_PART_SETS 8
int xeq_message_wq_edge_grad_sum(const void* vec, int64_t n_nodes, int64_t n_edges, const int32_t* qptr, const int32_t* peid,
                                 const int32_t* mirror, const int32_t mul[3], int n_sets, const void* const* parts, void* grad_vec,
                                 void* stream);

/* ------------------------------------------------- node-side fused elementwise stages
 * Internal "BT" layout of equivariant intermediates (xhat_layout = 1 above): block-major over
 * l, then node, then m, then channel: addr(n, u' in block l, m) = N*base_l + (n*(2l+1)+m)*W_l + u'
 * (W_l = mul_l, or 2 mul_l for the U|V pair buffer).  Each block is a plain row-major
 * [N(2l+1), W_l] matrix, so o3.Linear (nn/xpainn.py:186-187, 211-212) is three ordinary GEMMs. */

/* nn.LayerNorm(s) (nn/xpainn.py:123,130) + EquivariantLayerNorm(x) (nn/o3layer.py:145-171) in one
 * pass; shat has row stride ld_s, xhat is written in BT layout, stats[N,4] = (mean, rstd, mean0, r).
 * do_norm = 0: identity norms (layer_norm=False), still re-laying x out. */
int xeq_norm_fwd(int dtype, const void* s, const void* x, const void* ln_w, const void* ln_b, const void* eq_w,
                 const void* eq_b, int64_t n, int node_dim, const int32_t mul[3], int do_norm, void* shat, int64_t ld_s,
                 void* xhat_bt, void* stats, void* stream);
/* g_s = res_s + LN^T g_shat ; g_x = res_x + EqLN^T g_xhat_bt  (res_* may be NULL; g_x in e3nn layout). */
int xeq_norm_bwd(int dtype, const void* s, const void* x, const void* ln_w, const void* eq_w, const void* stats,
                 int64_t n, int node_dim, const int32_t mul[3], int do_norm, const void* g_shat, int64_t ld_gs,
                 const void* g_xhat_bt, const void* res_s, const void* res_x, void* g_s, void* g_x, void* stream);
/* Invariant(V) and EquivariantDot(U,V) (nn/xpainn.py:214,222; nn/o3layer.py:39-44,104-109) from the U|V
 * pair buffer: cat[n, node_dim + u] = sqrt(sum_m V^2 + eps^2) - eps, p[n,u] = sum_m U V; and the reverse. */
int xeq_uv_reduce_fwd(int dtype, const void* uv_bt, int64_t n, const int32_t mul[3], double eps, void* cat,
                      int64_t ld_cat, int node_dim, void* p, void* stream);
/* Reverse: g_U = dL/dU of the output stage + g_p V, g_V = g_p U + g_v V / (v + eps).  The output stage's part is formed
 * here as g_x_out[n,u,m] * a_vv[n,u] when g_x_out / a are given (then xeq_update_out_bwd is called with g_uv_bt = NULL);
 * with both NULL it is read from the U columns of g_uv_bt, where xeq_update_out_bwd left it. */
int xeq_uv_reduce_bwd(int dtype, const void* uv_bt, const void* g_p, const void* g_cat, int64_t ld_cat, int node_dim,
                      int64_t n, const int32_t mul[3], double eps, const void* g_x_out, const void* a, void* g_uv_bt,
                      void* stream);
/* Output stage of XPainnUpdate (nn/xpainn.py:218-229) with a = [a_vv C | a_sv F | a_ss F], ip = dot_lin(p):
 * s_out = s + a_sv*ip + a_ss, x_out = x + U (x) a_vv (x, x_out in e3nn layout; x_out may be NULL: not computed); and the reverse
 * (g_uv_bt may be NULL: dL/dU = g_x_out a_vv is then left to xeq_uv_reduce_bwd; g_x_out may be NULL: zero). */
int xeq_update_out_fwd(int dtype, const void* s, const void* x, const void* uv_bt, const void* a, const void* ip,
                       int64_t n, int node_dim, const int32_t mul[3], void* s_out, void* x_out, void* stream);
int xeq_update_out_bwd(int dtype, const void* g_s_out, const void* g_x_out, const void* uv_bt, const void* a,
                       const void* ip, int64_t n, int node_dim, const int32_t mul[3], void* g_a, void* g_ip,
                       void* g_uv_bt, void* stream);

/* ---- two-layer scalar MLPs on the matrix cores (f32, hidden width 128) ----------------------------
 * Replace nn.Sequential(Linear, SiLU, Linear) of XPainnMessage.scalar_mlp (nn/xpainn.py:103-107) and
 * XPainnUpdate.update_mlp (nn/xpainn.py:177-181) and their input gradients: one launch each, the hidden
 * activations stay on chip.  Weights are read from a packed copy in matrix-core fragment order with the bias as one
 * more k-group (xeq_mlp_pack; xeq_mlp_packed_floats(n_out, k_in) floats):
 *   packed W1  = pack(lin1.weight [128, k1], lin1.bias, transposed = 0)     forward stage 1
 *   packed W2  = pack(lin2.weight [n2, 128], lin2.bias, transposed = 0)     forward stage 2
 *   packed W2t = pack(lin2.weight viewed as [k_in = n2][n_out = 128], NULL, transposed = 1)   reverse stage 1
 *   packed W1t = pack(lin1.weight viewed as [k_in = 128][n_out = k1], NULL, transposed = 1)   reverse stage 2
 * xeq_mlp2_supported: 1 when the kernels take (dtype, k1, hidden, n2): f32, hidden 128, k1 % 32 == 0 (it is the
 * reverse pass's output width), n2 % 32 == 0. */
int xeq_mlp2_supported(int dtype, int k1, int hidden, int n2);
/* Work split of the node kernels (host logic): out = {tiles with one workgroup each, workgroups per remaining tile, grid size}
 * for `tiles` 32-node tiles whose second phase can be cut in at most `max_split` pieces. */
int xeq_node_tile_split(int64_t tiles, int max_split, int64_t out[3]);
int64_t xeq_mlp_packed_floats(int n_out, int k_in);
int xeq_mlp_pack(const float* w, const float* bias, int n_out, int k_in, int transposed, float* packed, void* stream);
/* y[n, n2] (row stride ldy, % 4 == 0) = silu(x[n, k1] (row stride ldx, % 4 == 0) W1^T + b1) W2^T + b2;
 * pre[n, 128] = the pre-activation, kept for the reverse pass. */
int xeq_mlp2_fwd(const float* x, int64_t ldx, int64_t n, int k1, const float* w1_packed, const float* w2_packed, int n2,
                 float* pre, float* y, int64_t ldy, void* stream);
/* gx[n, n2] = ((g[n, k1] W2) * silu'(pre)) W1 with k1 = the forward's n2 and n2 = the forward's k1. */
int xeq_mlp2_bwd(const float* g, int64_t ldg, int64_t n, int k1, const float* w2t_packed, const float* pre,
                 const float* w1t_packed, int n2, float* gx, int64_t ldgx, void* stream);
/* XPainnUpdate's two independent products side by side (reference nn/xpainn.py:219-223: a = update_mlp([shat | v]) and dot_lin(<U, V>);
 * reverse = 1: their input gradients): xeq_mlp2_fwd (reverse: xeq_mlp2_bwd with stage1_packed = w2t_packed, stage2_packed = w1t_packed)
 * and xeq_linear_fwd without bias / activation / row gather on the same n rows.  One launch when n takes the few-row forms
 * (XEQ_SMALL_ROWS, csrc/xeq_common.h), else the two launches in this order.  The same bits as the separate entry points. */
int xeq_mlp2_and_linear(int reverse, const float* x, int64_t ldx, int64_t n, int k1, const float* stage1_packed, const float* stage2_packed,
                        int n2, float* pre, float* y, int64_t ldy, const float* lin_x, int64_t lin_ldx, int lin_k,
                        const float* lin_w_packed, int lin_n_out, float* lin_y, int64_t lin_ldy, void* stream);

/* ---- first half of XPainnUpdate.forward in one launch (f32; mul_l in {0, 32, 64, 128}, node_dim <= 128, D <= 504) -------
 * nn.LayerNorm(s) and EquivariantLayerNorm(x) (nn/xpainn.py:208-209), U = update_U(xhat), V = update_V(xhat) (o3.Linear,
 * nn/xpainn.py:211-212), v = Invariant(V), p = EquivariantDot(U, V) (nn/xpainn.py:214, :222).  Replaces xeq_norm_fwd + the
 * three GEMMs + xeq_uv_reduce_fwd; outputs as theirs: cat[n, :node_dim] = shat, cat[n, node_dim:node_dim + C] = v (row stride
 * ld_cat), p[n, C], the U|V pair buffer uv_bt (BT layout), stats[n, 4].  w_packed_l = xeq_mlp_pack([W_U | W_V] / sqrt(mul_l)
 * viewed [k_in = mul_l][n_out = 2 mul_l], bias = [b_U | b_V] for l = 0 (has_bias) else NULL, transposed = 1); NULL for an
 * absent block.  xeq_update_uv_supported: 1 when the kernel takes the layout. */
int xeq_update_uv_supported(int dtype, int node_dim, const int32_t mul[3]);
int xeq_update_uv_fwd(const float* s, const float* x, const float* ln_w, const float* ln_b, const float* eq_w, const float* eq_b,
                      int64_t n, int node_dim, const int32_t mul[3], int do_norm, const float* w_packed0, const float* w_packed1,
                      const float* w_packed2, int has_bias, double eps, float* cat, int64_t ld_cat, float* p, float* uv_bt,
                      float* stats, void* stream);
/* Reverse of the above in one launch (replaces xeq_uv_reduce_bwd + three GEMMs + xeq_norm_bwd):
 *   g_U = g_x_out a_vv + g_p V,  g_V = g_p U + g_v V / sqrt(sum_m V^2 + eps^2)   (g_v = g_cat[:, node_dim:], a_vv = a[:, :C])
 *   g_xhat = g_U W_U^T + g_V W_V^T;  g_s = g_s_out + LN^T(g_cat[:, :node_dim]),  g_x = g_x_out + EqLN^T(g_xhat).
 * wt_packed_l = xeq_mlp_pack([W_U | W_V] / sqrt(mul_l) as [n_out = mul_l][k_in = 2 mul_l], NULL, transposed = 0).
 * g_x_out may be NULL (the block's equivariant output has no consumer: the last block of a force evaluation): zero.
 * Split form (g_xhat_bt != NULL): stops after the contraction and writes dL/dxhat in BT layout for xeq_norm_bwd (g_s_out, s,
 * x, stats, ln_w, eq_w, g_s, g_x unused; the workgroup then needs 50 KB of LDS instead of 113 KB). */
int xeq_update_uv_bwd(const float* uv_bt, const float* g_p, const float* g_cat, int64_t ld_cat, const float* g_x_out,
                      const float* g_s_out, const float* a, int64_t ld_a, const float* s, const float* x, const float* stats,
                      const float* ln_w, const float* eq_w, int64_t n, int node_dim, const int32_t mul[3], int do_norm,
                      const float* wt_packed0, const float* wt_packed1, const float* wt_packed2, double eps, float* g_s, float* g_x,
                      float* g_xhat_bt, void* stream);

/* Epoch of the packed weight copies: every pack cache (nn/fused.py, nn/nodeblock.py, csrc/xeq_torch.cpp) keys on it next to the version
 * counters of the tensors it packed; what changes parameters without bumping those counters (a replayed captured optimizer step,
 * train.GraphedTrainStep) calls xeq_pack_epoch_bump and the next evaluation repacks.  Host only. */
long long xeq_pack_epoch(void);
void xeq_pack_epoch_bump(void);
/* Capacity form of the open-boundary list: count[0] = raw[n_nodes] (the true edge count); rowptr = raw when the list fits `capacity`,
 * all zeros (an EMPTY list) when it does not -- a cut list would not be symmetric and the symmetric shortcuts downstream index by the
 * reverse edge (replaces the unguarded row pointer of round 3: a list that outgrew its arrays now leads no kernel past a buffer). */
int xeq_rowptr_guard(const int32_t* raw, int64_t n_nodes, int64_t capacity, int32_t* rowptr, int32_t* count, void* stream);
/* Degrees -> guarded row pointer in ONE launch (round 5): rowptr[0 .. n] = exclusive prefix sums of deg[0 .. n) with
 * xeq_rowptr_guard's rule (a list beyond `capacity` becomes EMPTY; capacity < 0: no guard), count[0] (optional) = the true total,
 * running_total[0] (optional, int64) += the true total: a device-side counter of the edges a replayed step has processed.
 * One workgroup; n_nodes <= xeq_rowptr_from_degrees_max() (XEQ_ERR_UNSUPPORTED above: xeq_exclusive_scan_i32_ws + xeq_rowptr_guard). */
int xeq_rowptr_from_degrees(const int32_t* deg, int64_t n_nodes, int64_t capacity, int32_t* rowptr, int32_t* count, int64_t* running_total,
                            void* stream);
int64_t xeq_rowptr_from_degrees_max(void);   /* a size, not a status */

/* ---- the per-node chain between two message aggregations as ONE launch per direction (round 4; csrc/xeq_nodeblock.hip) -------------
 * f32, the default layout (node_dim 128, 128x0e + 64x1o + 32x2e: xeq_node_block_supported).  A wave owns 16 nodes and keeps their
 * activations in the matrix cores' accumulator layout; every contraction runs on bf16 MFMAs (16x16x32) over three-way split operands
 * (six products per k-step, f32 accumulation); the weights stream through LDS from a copy packed in consumption order.
 *
 * xeq_node_block_fwd = XPainnUpdate.forward (nn/xpainn.py:206-231: both norms, o3.Linear U / V, Invariant, EquivariantDot, update_mlp,
 * dot_lin, residual update) and, when h_next != NULL, the front half of the NEXT XPainnMessage.forward (nn/xpainn.py:128-139: both
 * norms of (s_out, x_out) and scalar_mlp).  It replaces xeq_update_uv_fwd + xeq_mlp2_fwd + xeq_linear_fwd + xeq_update_out_fwd
 * (+ xeq_norm_fwd + xeq_mlp2_fwd of the next block).  What other kernels read keeps their layouts: s_out [n, F], x_out [n, D] (NULL: no
 * consumer), stats [n, 4], and for the next block stats_next [n, 4], xhat_next (BT), h_next [n, F + 2 C].  What only the reverse launch
 * reads is INTERNAL: uv (U|V), pre (update_mlp hidden pre-activation), a (update_mlp output), ip (dot_lin output), pre_next and the
 * scratch p -- xeq_node_block_rows(n) rows each (whole workgroups of 64 nodes) in the wave-native layout [block of 16 nodes][tile of
 * 32 channels][register quad g][lane = 16 q + node][4 floats] (channel 16 g + 4 q + e of the tile), in which every wave access is 1 KB of consecutive bytes (tile numbering:
 * csrc/xeq_nodeblock.hip, uv_tile / x_tile; nn/nodeblock.py::native_to_rows turns one into rows).
 * packed: xeq_node_block_pack_fwd(update_mlp[0].weight [F, F + C], [W_U | W_V] / sqrt(mul_l) as [mul_l, 2 mul_l] for l = 0, 1, 2,
 * dot_lin.weight [F, C], update_mlp[2].weight [C + 2 F, F], next scalar_mlp[0].weight [F, F] and [2].weight [F + 2 C, F] (both
 * NULL: without the next block), out, stream); out holds xeq_node_block_fwd_tiles(with_tail) * 3072 bytes.
 * b_uv = [update_U.bias | update_V.bias] ([2 F]) or NULL; p_scratch: [n, C] floats (EquivariantDot(U, V), read back by dot_lin). */
int xeq_node_block_supported(int dtype, int node_dim, const int32_t mul[3]);
/* launch policy (host only): 1 when an evaluation of n nodes should take the fused launches (n >= XEQ_NODE_BLOCK_MIN_NODES, default
 * 6 144; XEQ_NODE_BLOCK=0: never).  Below that the chain of small kernels is faster (a fused launch is one serial chain per wave). */
int xeq_node_block_auto(int64_t n);
/* launch policy (host only): the row count up to which xeq_linear_fwd, xeq_mlp2_fwd / _bwd, xeq_mlp2_and_linear and xeq_update_uv_fwd /
 * _bwd (split form) take their few-row forms -- 16 x 16 exact-f32 tiles whose results equal the 32-row kernels' bit for bit (MD-sized
 * systems: a quarter of the k-chain per wave, four times the waves).  3 584; XEQ_SMALL_ROWS overrides (0: never), read per call. */
int64_t xeq_small_rows_limit(void);
/* waves per workgroup of the node-block launches that follow (host only, process-wide): 0 = by node count (one workgroup per CU of 5 .. 8
 * waves between 4 097 and 8 192 wave-blocks of 16 nodes, four otherwise: a lone launch ends with its slowest CU), 4 .. 8 = fixed.  Results
 * do not depend on it.  -> the former setting; -1: out of range, nothing changed. */
int xeq_node_block_set_waves(int waves);
int64_t xeq_node_block_rows(int64_t n);
int64_t xeq_node_block_fwd_tiles(int with_tail);
int xeq_node_block_pack_fwd(const float* w3, const float* uv0, const float* uv1, const float* uv2, const float* dot, const float* w4,
                            const float* w1_next, const float* w2_next, void* out, void* stream);
int xeq_node_block_fwd(int64_t n, const float* s, const float* x, const float* ln_w, const float* ln_b, const float* eq_w, const float* eq_b,
                       const float* b_uv, const float* b3, const float* b4, double eps, const void* packed, float* p_scratch, float* uv_bt, float* stats,
                       float* pre, float* a, float* ip, float* s_out, float* x_out, const float* ln_w_next, const float* ln_b_next,
                       const float* eq_w_next, const float* eq_b_next, const float* b1_next, const float* b2_next, float* stats_next,
                       float* xhat_next, float* pre_next, float* h_next, void* stream);
/* Reverse of xeq_node_block_fwd for a force evaluation (input gradients only, nn/basic.py:143-159): replaces xeq_mlp2_bwd + xeq_norm_bwd
 * of the next message block and xeq_update_out_bwd + xeq_linear_fwd (dot_lin^T) + xeq_mlp2_bwd + xeq_update_uv_bwd + xeq_norm_bwd of the
 * update block.  With the next block's front half (g_h != NULL): g_h [n, F + 2 C] and g_xhat_next (BT) are the gradients of h_next /
 * xhat_next, g_s_in / g_x_in the gradients that reach s_out / x_out directly (the message kernel's residual path), and s_out, x_out,
 * stats_next, pre_next what the forward launch wrote.  Without it: g_s_in = dL/ds_out, g_x_in = dL/dx_out or NULL (zero: the last
 * block of a force evaluation; packed with with_gx = 0).  Scratch (caller-owned, xeq_node_block_rows(n) rows each, internal layout):
 * gxo [., D] (whenever dL/dx_out is not zero), gp [., C], gv [., C], gw [., D].  Output: g_s [n, F], g_x [n, D].  packed: xeq_node_block_pack_bwd of the same weight tensors as the forward pack. */
int64_t xeq_node_block_bwd_tiles(int with_tail, int with_gx);
int xeq_node_block_pack_bwd(const float* w3, const float* uv0, const float* uv1, const float* uv2, const float* dot, const float* w4,
                            const float* w1_next, const float* w2_next, int with_gx, void* out, void* stream);
int xeq_node_block_bwd(int64_t n, const float* g_h, const float* g_xhat_next, const float* g_s_in, const float* g_x_in, const float* s_out,
                       const float* x_out, const float* stats_next, const float* pre_next, const float* ln_w_next, const float* eq_w_next,
                       const float* uv_bt, const float* a, const float* ip, const float* pre, const float* s, const float* x,
                       const float* stats, const float* ln_w, const float* eq_w, double eps, const void* packed, float* gxo, float* gp,
                       float* gv, float* gw, float* g_s, float* g_x, void* stream);
/* test entry: y [n, 32 n_ot] = x [n, 128] W^T (W [32 n_ot, 128]) through the kernel's primitives; form 0 (n_ot = 4): chunk
 * accumulation, form 1: one output tile at a time; packed_scratch: 8 n_ot * 3072 bytes */
int xeq_node_block_linear_test(const float* x, int64_t n, const float* w, int n_ot, int form, void* packed_scratch, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XEQ_H */
