// TorchScript-visible operators over the C ABI of libxeq_hip.so (include/xeq.h).
//
// Reference: the reference ships its models to LAMMPS / GROMACS as TorchScript files (run/jit_script.py:28-86 scripts
// interface/jit_model.py:12-216 and saves it with `_extra_files`); its hot ops live in third-party extensions that are
// themselves registered torch operators (torch_scatter, torch_cluster), so `torch.jit.script` sees schemas, not Python.
// This file gives the HIP path the same footing: operators in the `xeq::` namespace with schemas, registered through
// TORCH_LIBRARY, loadable from Python (`torch.ops.load_library`) and from a libtorch host program (dlopen before
// torch::jit::load).
//
//   xeq::xpainn_eval      ONE operator for a whole energy (+ forces, + virial) evaluation of XPaiNN (nn/model.py:26-46):
//                         edge geometry -> embedding -> n x [message + update] -> energy head -> explicit reverse pass.
//                         Every stage is the same HIP kernel the Python modules launch (nn/fused.py is the Python twin of
//                         this file and the two are compared bit for bit in tests/test_gpu_interface.py); the scalar MLPs
//                         are xeq_mlp2_fwd / _bwd, the o3.Linear contractions ATen GEMMs.  Enqueued from C++: a batch with a never-seen topology costs its
//                         GPU time plus ~150 native launches, no Python between kernels and no graph capture.
//                         Autograd: `energy` is differentiable w.r.t. `pos` (backward = -forces), which is how the
//                         GROMACS-style model hands forces to its caller (interface/jit_model.py:208-214).
//   xeq::radius_graph     open-boundary neighbour list (data/transform.py:58-64), canonical (center, neighbor) order.
//   xeq::radius_graph_pbc periodic single-system neighbour list with the reference's order and cell offsets
//                         (data/radius_graph.py:195-275 `single_radius_graph`, called inside the scripted GROMACS model,
//                         interface/jit_model.py:183-195).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <optional>
#include <unordered_map>
#include <vector>

#include "../../include/xeq.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

#define XCALL(...)                                                                                  \
  do {                                                                                              \
    const int st_ = (__VA_ARGS__);                                                                  \
    TORCH_CHECK(st_ == 0, "xequinet_amd: ", #__VA_ARGS__, " failed (", st_, "): ", xeq_last_error()); \
  } while (0)

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
const void* P(const Tensor& t) { return t.defined() && t.numel() > 0 ? t.data_ptr() : (t.defined() ? t.data_ptr() : nullptr); }
void* Pm(Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
int dcode(const Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kDouble, "xequinet_amd: float32 / float64 tensors only");
  return t.scalar_type() == at::kFloat ? XEQ_F32 : XEQ_F64;
}
void need_hip(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "xequinet_amd ops run on MI355X (HIP) tensors only and have no CPU fallback; ", what, " is on ", t.device());
}

// ---------------------------------------------------------------------------------------------- two-layer MLPs
// Linear-SiLU-Linear on the matrix cores (xeq_mlp2_fwd / _bwd, csrc/xeq_mlp.hip); nn/fused.py::_mlp_fwd / _mlp_bwd are the
// Python twins.  The fragment-order weight copies are cached per weight tensor and rebuilt when a version counter moves.
// An entry is valid only while the storage it was packed from is ALIVE (weak references): a freed parameter's address is
// reused by the allocator, typically by the next model's parameter of the same shape with the same version count.
// The references are to the STORAGE (not the tensor object): XPaiNNNative hands over fresh views of one flat parameter
// buffer on every call -- same storage, same address, shared version counter -- and those must hit the cache.
using WeakStore = c10::weak_intrusive_ptr<c10::StorageImpl>;
struct Owners {
  std::vector<std::optional<WeakStore>> refs;   // nullopt: an undefined tensor (absent bias)
  void set(std::initializer_list<const Tensor*> ts) {
    refs.clear();
    for (const Tensor* t : ts) {
      if (t->defined() && t->has_storage()) refs.emplace_back(t->storage().getWeakStorageImpl());
      else refs.emplace_back(std::nullopt);
    }
  }
  bool same(std::initializer_list<const Tensor*> ts) const {
    if (refs.size() != ts.size()) return false;
    size_t i = 0;
    for (const Tensor* t : ts) {
      const bool d = t->defined() && t->has_storage();
      if (d != refs[i].has_value()) return false;
      if (d) {
        auto sp = refs[i]->lock();
        if (!sp || sp.get() != t->storage().unsafeGetStorageImpl()) return false;
      }
      ++i;
    }
    return true;
  }
};
struct MlpPacks {
  int64_t key[8];
  Owners owners;
  Tensor w1p, w2p, w2tp, w1tp;
};
const MlpPacks* mlp_packs(const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2) {
  if (w1.scalar_type() != at::kFloat || b1.numel() == 0 || b2.numel() == 0 ||
      !xeq_mlp2_supported(XEQ_F32, (int)w1.size(1), (int)w1.size(0), (int)w2.size(0)))
    return nullptr;
  static std::mutex mu;
  static std::unordered_map<const void*, MlpPacks> cache;
  const int64_t key[8] = {(int64_t)w1._version() + ((int64_t)xeq_pack_epoch() << 32), (int64_t)(intptr_t)w1.data_ptr(), (int64_t)b1._version(), (int64_t)(intptr_t)b1.data_ptr(),
                          (int64_t)w2._version(), (int64_t)(intptr_t)w2.data_ptr(), (int64_t)b2._version(), (int64_t)(intptr_t)b2.data_ptr()};
  std::lock_guard<std::mutex> lock(mu);
  MlpPacks& e = cache[w1.data_ptr()];
  bool same = e.w1p.defined() && e.owners.same({&w1, &b1, &w2, &b2});
  for (int i = 0; i < 8 && same; ++i) same = e.key[i] == key[i];
  if (!same) {
    const int h = (int)w1.size(0), k1 = (int)w1.size(1), n2 = (int)w2.size(0);
    const Tensor w1c = w1.detach().contiguous(), w2c = w2.detach().contiguous();
    auto pack = [&](const Tensor& w, const Tensor* bias, int n_out, int k_in, int transposed) {
      Tensor out = at::empty({xeq_mlp_packed_floats(n_out, k_in)}, w.options());
      XCALL(xeq_mlp_pack((const float*)w.data_ptr(), bias ? (const float*)bias->data_ptr() : nullptr, n_out, k_in, transposed,
                         (float*)out.data_ptr(), cur_stream()));
      return out;
    };
    e.w1p = pack(w1c, &b1, h, k1, 0);
    e.w2p = pack(w2c, &b2, n2, h, 0);
    e.w2tp = pack(w2c, nullptr, h, n2, 1);
    e.w1tp = pack(w1c, nullptr, k1, h, 1);
    e.owners.set({&w1, &b1, &w2, &b2});
    for (int i = 0; i < 8; ++i) e.key[i] = key[i];
  }
  return &e;
}
// rbf_lin's rows in the wq message kernels' LDS layout (xeq_message_wq_pack_weights; ops.wq_packed_weights is the Python twin), cached
// per weight version like the packs above: the kernels then stage a unit's weights with coalesced loads (XEQ_WQ_PACKED_WEIGHTS)
struct WqWeightPack {
  int64_t key[4];
  Owners owners;
  Tensor packed;
};
const Tensor* wq_weight_pack(const Tensor& w, const Tensor& b, int num_basis, int node_dim, const int32_t mul[3]) {
  const int64_t n = xeq_message_wq_packed_weight_floats(num_basis, node_dim, mul);
  if (n <= 0 || w.scalar_type() != at::kFloat || !b.defined() || b.numel() == 0) return nullptr;
  static std::mutex mu;
  static std::unordered_map<const void*, WqWeightPack> cache;
  const int64_t key[4] = {(int64_t)w._version() + ((int64_t)xeq_pack_epoch() << 32), (int64_t)(intptr_t)w.data_ptr(), (int64_t)b._version(), (int64_t)(intptr_t)b.data_ptr()};
  std::lock_guard<std::mutex> lock(mu);
  WqWeightPack& e = cache[w.data_ptr()];
  bool same = e.packed.defined() && e.owners.same({&w, &b});
  for (int i = 0; i < 4 && same; ++i) same = e.key[i] == key[i];
  if (!same) {
    const Tensor wc = w.detach().contiguous(), bc = b.detach().contiguous();
    e.packed = at::empty({n}, w.options());
    XCALL(xeq_message_wq_pack_weights(wc.data_ptr(), bc.data_ptr(), num_basis, node_dim, mul, e.packed.data_ptr(), cur_stream()));
    e.owners.set({&w, &b});
    for (int i = 0; i < 4; ++i) e.key[i] = key[i];
  }
  return &e.packed;
}
// x: [n, k1] rows with stride ldx (a column slice of a wider buffer is fine)
void mlp_fwd(const Tensor& x, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, Tensor& pre, Tensor& y) {
  const MlpPacks* pk = (x.stride(1) == 1 && x.stride(0) % 4 == 0) ? mlp_packs(w1, b1, w2, b2) : nullptr;
  if (!pk) {
    pre = at::addmm(b1, x, w1.t());
    y = at::addmm(b2, at::silu(pre), w2.t());
    return;
  }
  const int64_t n = x.size(0);
  pre = at::empty({n, w1.size(0)}, x.options());
  y = at::empty({n, w2.size(0)}, x.options());
  XCALL(xeq_mlp2_fwd((const float*)x.data_ptr(), x.stride(0), n, (int)w1.size(1), (const float*)pk->w1p.data_ptr(),
                     (const float*)pk->w2p.data_ptr(), (int)w2.size(0), (float*)pre.data_ptr(), (float*)y.data_ptr(), w2.size(0), cur_stream()));
}
Tensor mlp_bwd(const Tensor& g_y, const Tensor& pre, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2) {
  const MlpPacks* pk = mlp_packs(w1, b1, w2, b2);
  if (!pk) return at::mm(at::silu_backward(at::mm(g_y, w2), pre), w1);
  const Tensor g = g_y.contiguous();
  const int64_t n = g.size(0);
  Tensor g_x = at::empty({n, w1.size(1)}, g.options());
  XCALL(xeq_mlp2_bwd((const float*)g.data_ptr(), g.size(1), n, (int)g.size(1), (const float*)pk->w2tp.data_ptr(), (const float*)pre.data_ptr(),
                     (const float*)pk->w1tp.data_ptr(), (int)w1.size(1), (float*)g_x.data_ptr(), w1.size(1), cur_stream()));
  return g_x;
}

// Single linear layers (dot_lin, the embedding, the head's first layer) through xeq_linear_fwd (csrc/xeq_linear.hip);
// nn/fused.py::_linear_pack / _linear / linear_module_fwd / linear_module_bwd are the Python twins.  Packs cached per weight.
struct LinPack {
  int64_t key[4];
  Owners owners;
  Tensor fwd, bwd;
};
const LinPack* lin_pack(const Tensor& w, const Tensor& b) {
  const int n_out = (int)w.size(0), k_in = (int)w.size(1);
  if (w.scalar_type() != at::kFloat || !xeq_linear_supported(XEQ_F32, k_in, n_out)) return nullptr;
  const bool has_bwd = xeq_linear_supported(XEQ_F32, n_out, k_in) != 0;   // (the embedding's 56 inputs: forward only, nobody differentiates it)
  static std::mutex mu;
  static std::unordered_map<const void*, LinPack> cache;
  const bool hb = b.defined() && b.numel() > 0;
  const int64_t key[4] = {(int64_t)w._version() + ((int64_t)xeq_pack_epoch() << 32), (int64_t)(intptr_t)w.data_ptr(), hb ? (int64_t)b._version() : -1, hb ? (int64_t)(intptr_t)b.data_ptr() : 0};
  std::lock_guard<std::mutex> lock(mu);
  LinPack& e = cache[w.data_ptr()];
  bool same = e.fwd.defined() && e.owners.same({&w, &b});
  for (int i = 0; i < 4 && same; ++i) same = e.key[i] == key[i];
  if (!same) {
    const Tensor wc = w.detach().contiguous();
    e.fwd = at::empty({xeq_mlp_packed_floats(n_out, k_in)}, w.options());
    XCALL(xeq_mlp_pack((const float*)wc.data_ptr(), hb ? (const float*)b.data_ptr() : nullptr, n_out, k_in, 0, (float*)e.fwd.data_ptr(), cur_stream()));
    e.bwd = Tensor();
    if (has_bwd) {
      e.bwd = at::empty({xeq_mlp_packed_floats(k_in, n_out)}, w.options());
      XCALL(xeq_mlp_pack((const float*)wc.data_ptr(), nullptr, k_in, n_out, 1, (float*)e.bwd.data_ptr(), cur_stream()));
    }
    e.owners.set({&w, &b});
    for (int i = 0; i < 4; ++i) e.key[i] = key[i];
  }
  return &e;
}
// y = act(x W^T + b); row_index (int32, optional) gathers the rows of x; pre (optional) receives the pre-activation
Tensor linear_fwd(const Tensor& x, const Tensor& w, const Tensor& b, int act = 0, const Tensor* row_index = nullptr, Tensor* pre = nullptr) {
  const bool hb = b.defined() && b.numel() > 0;
  const LinPack* pk = (x.dim() == 2 && x.stride(1) == 1 && x.stride(0) % 4 == 0) ? lin_pack(w, b) : nullptr;
  if (!pk) {
    const Tensor xr = row_index ? x.index_select(0, row_index->to(at::kLong)) : x;
    Tensor y = hb ? at::addmm(b, xr, w.t()) : at::mm(xr, w.t());
    if (pre) *pre = y;
    return act == 1 ? at::silu(y) : y;
  }
  const int64_t n = row_index ? row_index->numel() : x.size(0);
  Tensor y = at::empty({n, w.size(0)}, x.options());
  if (pre) *pre = at::empty({n, w.size(0)}, x.options());
  XCALL(xeq_linear_fwd(x.data_ptr(), x.stride(0), n, (int)w.size(1), row_index ? (const int32_t*)row_index->data_ptr() : nullptr,
                       pk->fwd.data_ptr(), (int)w.size(0), hb, act, pre ? pre->data_ptr() : nullptr, y.data_ptr(), w.size(0), cur_stream()));
  return y;
}
Tensor linear_bwd(const Tensor& g_in, const Tensor& w, const Tensor& b) {   // dL/dx = g W
  const LinPack* pk = lin_pack(w, b);
  if (!pk || !pk->bwd.defined()) return at::mm(g_in, w);
  const Tensor g = g_in.contiguous();
  Tensor gx = at::empty({g.size(0), w.size(1)}, g.options());
  XCALL(xeq_linear_fwd(g.data_ptr(), g.stride(0), g.size(0), (int)w.size(0), nullptr, pk->bwd.data_ptr(), (int)w.size(1), 0, 0, nullptr,
                       gx.data_ptr(), w.size(1), cur_stream()));
  return gx;
}

// XPainnUpdate's two independent products side by side (xeq_mlp2_and_linear: one launch for MD-sized systems);
// nn/fused.py::mlp_and_linear_fwd / _bwd are the Python twins.
void mlp_and_linear_fwd(const Tensor& x, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, const Tensor& p, const Tensor& wl,
                        Tensor& pre, Tensor& y, Tensor& ip) {
  const MlpPacks* pk = (x.stride(1) == 1 && x.stride(0) % 4 == 0) ? mlp_packs(w1, b1, w2, b2) : nullptr;
  const LinPack* lp = (p.dim() == 2 && p.stride(1) == 1 && p.stride(0) % 4 == 0) ? lin_pack(wl, Tensor()) : nullptr;
  if (!pk || !lp) {
    mlp_fwd(x, w1, b1, w2, b2, pre, y);
    ip = linear_fwd(p, wl, Tensor());
    return;
  }
  const int64_t n = x.size(0);
  pre = at::empty({n, w1.size(0)}, x.options());
  y = at::empty({n, w2.size(0)}, x.options());
  ip = at::empty({n, wl.size(0)}, x.options());
  XCALL(xeq_mlp2_and_linear(0, (const float*)x.data_ptr(), x.stride(0), n, (int)w1.size(1), (const float*)pk->w1p.data_ptr(),
                            (const float*)pk->w2p.data_ptr(), (int)w2.size(0), (float*)pre.data_ptr(), (float*)y.data_ptr(), w2.size(0),
                            (const float*)p.data_ptr(), p.stride(0), (int)wl.size(1), (const float*)lp->fwd.data_ptr(), (int)wl.size(0),
                            (float*)ip.data_ptr(), wl.size(0), cur_stream()));
}
void mlp_and_linear_bwd(const Tensor& g_y, const Tensor& pre, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2,
                        const Tensor& g_lin_in, const Tensor& wl, Tensor& g_x, Tensor& g_p) {
  const MlpPacks* pk = mlp_packs(w1, b1, w2, b2);
  const LinPack* lp = lin_pack(wl, Tensor());
  if (!pk || !lp || !lp->bwd.defined()) {
    g_p = linear_bwd(g_lin_in, wl, Tensor());
    g_x = mlp_bwd(g_y, pre, w1, b1, w2, b2);
    return;
  }
  const Tensor g = g_y.contiguous(), gl = g_lin_in.contiguous();
  const int64_t n = g.size(0);
  g_x = at::empty({n, w1.size(1)}, g.options());
  g_p = at::empty({n, wl.size(1)}, g.options());
  XCALL(xeq_mlp2_and_linear(1, (const float*)g.data_ptr(), g.size(1), n, (int)g.size(1), (const float*)pk->w2tp.data_ptr(),
                            (const float*)pk->w1tp.data_ptr(), (int)w1.size(1), (float*)pre.data_ptr(), (float*)g_x.data_ptr(), w1.size(1),
                            (const float*)gl.data_ptr(), gl.stride(0), (int)wl.size(0), (const float*)lp->bwd.data_ptr(), (int)wl.size(1),
                            (float*)g_p.data_ptr(), wl.size(1), cur_stream()));
}

// [W_U | W_V] / sqrt(mul) blocks (prm layout: one [mul, 2 mul] tensor per l, empty when absent) in fragment order for
// xeq_update_uv_fwd; nn/fused.py::_packed_uv_frag is the Python twin.  Cached per weight tensor like the MLP packs.
struct UvFrag {
  int64_t key[8];
  Owners owners;
  Tensor w[3], wt[3];   // forward ([k_in = mul][n_out = 2 mul], biases folded) and reverse ([n_out = mul][k_in = 2 mul]) packs
};
constexpr int64_t UV_BWD_FUSE_NORM_MAX_NODES = 0;   // nn/fused.py::UV_BWD_FUSE_NORM_MAX_NODES (never fused: batch-independent bits)
const UvFrag* uv_frag(const Tensor* q /* [W0, W1, W2, bias pair] */, int node_dim, const int32_t mul[3]) {
  if (q[0].scalar_type() != at::kFloat || !xeq_update_uv_supported(XEQ_F32, node_dim, mul)) return nullptr;
  static std::mutex mu;
  static std::unordered_map<const void*, UvFrag> cache;
  int64_t key[8];
  for (int i = 0; i < 4; ++i) {
    key[2 * i] = q[i].numel() > 0 ? (int64_t)q[i]._version() + ((int64_t)xeq_pack_epoch() << 32) : -1;
    key[2 * i + 1] = q[i].numel() > 0 ? (int64_t)(intptr_t)q[i].data_ptr() : 0;
  }
  std::lock_guard<std::mutex> lock(mu);
  int first = 0;
  while (first < 3 && mul[first] == 0) ++first;
  UvFrag& e = cache[q[first].data_ptr()];
  bool same = e.w[first].defined() && e.owners.same({&q[0], &q[1], &q[2], &q[3]});
  for (int i = 0; i < 8 && same; ++i) same = e.key[i] == key[i];
  if (!same) {
    for (int l = 0; l < 3; ++l) {
      e.w[l] = Tensor();
      e.wt[l] = Tensor();
      if (mul[l] == 0) continue;
      const Tensor W = q[l].detach().contiguous();   // [k_in = mul][n_out = 2 mul]
      e.w[l] = at::empty({xeq_mlp_packed_floats(2 * mul[l], mul[l])}, W.options());
      XCALL(xeq_mlp_pack((const float*)W.data_ptr(), (l == 0 && q[3].numel() > 0) ? (const float*)q[3].data_ptr() : nullptr, 2 * mul[l],
                         mul[l], 1, (float*)e.w[l].data_ptr(), cur_stream()));
      e.wt[l] = at::empty({xeq_mlp_packed_floats(mul[l], 2 * mul[l])}, W.options());
      XCALL(xeq_mlp_pack((const float*)W.data_ptr(), nullptr, mul[l], 2 * mul[l], 0, (float*)e.wt[l].data_ptr(), cur_stream()));
    }
    e.owners.set({&q[0], &q[1], &q[2], &q[3]});
    for (int i = 0; i < 8; ++i) e.key[i] = key[i];
  }
  return &e;
}

// ---------------------------------------------------------------------------------------------- fused node blocks
// One launch per direction for an update block and the front half of the message block behind it (csrc/xeq_nodeblock.hip;
// nn/nodeblock.py / nn/fused.py::NodeBlock are the Python twins).  The packed weight programs are cached per update_mlp weight.
struct NbPacks {
  std::vector<int64_t> key;
  Owners owners;
  Tensor fwd, bwd, bias_uv;
};
// q: the block's parameters (layout below), qn: the next block's (nullptr: no front half); gx: dL/dx_out is not zero
const NbPacks* nb_packs(const Tensor* q, const Tensor* qn, bool gx) {
  static std::mutex mu;
  static std::unordered_map<const void*, NbPacks> cache[4];
  const bool tail = qn != nullptr;
  std::vector<const Tensor*> ws = {&q[15], &q[10], &q[11], &q[12], &q[14], &q[17], &q[13]};
  if (tail) {
    ws.push_back(&qn[0]);
    ws.push_back(&qn[2]);
  }
  std::vector<int64_t> key;
  for (const Tensor* t : ws) {
    key.push_back(t->defined() ? (int64_t)t->_version() + ((int64_t)xeq_pack_epoch() << 32) : -1);
    key.push_back(t->defined() && t->numel() > 0 ? (int64_t)(intptr_t)t->data_ptr() : 0);
  }
  std::lock_guard<std::mutex> lock(mu);
  NbPacks& e = cache[(tail ? 2 : 0) + (gx ? 1 : 0)][q[15].data_ptr()];
  bool same = e.fwd.defined() && e.key == key;
  if (same) {
    if (tail) same = e.owners.same({ws[0], ws[1], ws[2], ws[3], ws[4], ws[5], ws[6], ws[7], ws[8]});
    else same = e.owners.same({ws[0], ws[1], ws[2], ws[3], ws[4], ws[5], ws[6]});
  }
  if (!same) {
    const auto bopt = q[15].options().dtype(at::kByte);
    const Tensor w3 = q[15].detach().contiguous(), dot = q[14].detach().contiguous(), w4 = q[17].detach().contiguous();
    const Tensor uv0 = q[10].contiguous(), uv1 = q[11].contiguous(), uv2 = q[12].contiguous();
    Tensor w1n, w2n;
    if (tail) {
      w1n = qn[0].detach().contiguous();
      w2n = qn[2].detach().contiguous();
    }
    auto fp = [](const Tensor& t) { return t.defined() ? (const float*)t.data_ptr() : nullptr; };
    e.fwd = at::empty({xeq_node_block_fwd_tiles(tail) * 3072}, bopt);
    XCALL(xeq_node_block_pack_fwd(fp(w3), fp(uv0), fp(uv1), fp(uv2), fp(dot), fp(w4), fp(w1n), fp(w2n), e.fwd.data_ptr(), cur_stream()));
    e.bwd = at::empty({xeq_node_block_bwd_tiles(tail, gx) * 3072}, bopt);
    XCALL(xeq_node_block_pack_bwd(fp(w3), fp(uv0), fp(uv1), fp(uv2), fp(dot), fp(w4), fp(w1n), fp(w2n), gx, e.bwd.data_ptr(), cur_stream()));
    e.bias_uv = q[13].numel() > 0 ? q[13].detach().contiguous() : Tensor();
    if (tail) e.owners.set({ws[0], ws[1], ws[2], ws[3], ws[4], ws[5], ws[6], ws[7], ws[8]});
    else e.owners.set({ws[0], ws[1], ws[2], ws[3], ws[4], ws[5], ws[6]});
    e.key = key;
  }
  return &e;
}

// ---------------------------------------------------------------------------------------------- graph plumbing
struct WqPlan {
  int n_ranges = 0;
  int64_t pcap = 0;
  Tensor qptr, pgath, peid, qinfo, sq, sn, win, rowptr, work, basis, dbasis;
};
struct Graph {
  int64_t N = 0, E = 0;
  bool mirror = false;   // symmetric center-sorted list: the reverse wq kernel walks the forward plan (XEQ_WQ_MIRROR_WALK), ops.EdgeGraph.mirror_walk
  Tensor ei, c_rowptr, c_perm, n_rowptr, n_perm;   // c_perm undefined: edges already center-sorted.  n_*: the neighbor-sorted view (sorted_view)
  Tensor mirror_map;     // position of every edge's mirror edge (-1: none), for a list with `mirror`; an open list's is a permutation (= n_perm)
  WqPlan fwd, rev;
  Tensor sb_basis, sb_dbasis;
  void sorted_view();    // builds n_rowptr / n_perm by a stable sort when nobody has yet (a periodic mirror map is not a permutation)
  // the edges by neighbor for a kernel that only sums over them (xeq_edge_vectors_bwd): the mirror map over the center rows, else the sorted view
  const Tensor& rev_rowptr() { if (!mirror_map.defined()) sorted_view(); return mirror_map.defined() ? c_rowptr : n_rowptr; }
  const Tensor& rev_perm() { if (!mirror_map.defined()) sorted_view(); return mirror_map.defined() ? mirror_map : n_perm; }
};

Tensor i32(int64_t n, const Tensor& like) { return at::empty({std::max<int64_t>(n, 1)}, like.options().dtype(at::kInt)); }

void csr_by_key(const Tensor& keys, int64_t n_rows, Tensor& rowptr, Tensor& perm) {
  const int64_t n = keys.numel(), bytes = xeq_csr_by_key_workspace(n, n_rows);
  TORCH_CHECK(bytes >= 0, "xequinet_amd: csr_by_key: sizes out of range");
  Tensor work = at::empty({std::max<int64_t>(bytes, 1)}, keys.options().dtype(at::kByte));
  rowptr = i32(n_rows + 1, keys);
  perm = i32(n, keys);
  XCALL(xeq_csr_by_key((const int64_t*)keys.data_ptr(), n, n_rows, work.data_ptr(), bytes, (int32_t*)rowptr.data_ptr(),
                       (int32_t*)perm.data_ptr(), cur_stream()));
}

void Graph::sorted_view() {
  if (!n_perm.defined()) csr_by_key(ei.select(0, 1), N, n_rowptr, n_perm);
}

// ops.EdgeGraph's twin.  symmetric && center_sorted: the promise of this package's list builders -- open boundaries: neighbours ascending
// and unique per center, (i, j) present iff (j, i) is; with cell_offsets (a periodic list): a center's edges ascending in (neighbor, image),
// (i, j, o) present iff (j, i, -o) is up to a rounding at the cutoff (include/xeq.h, xeq_reverse_edge_map_pbc).
Graph build_graph(const Tensor& edge_index, int64_t n_nodes, bool center_sorted, bool symmetric, const Tensor* cell_offsets) {
  Graph g;
  g.N = n_nodes;
  g.ei = edge_index.contiguous();
  g.E = g.ei.size(1);
  const Tensor center = g.ei.select(0, 0), nbr = g.ei.select(0, 1);
  if (center_sorted) {
    g.c_rowptr = i32(n_nodes + 1, g.ei);
    XCALL(xeq_csr_rowptr((const int64_t*)center.data_ptr(), g.E, n_nodes, (int32_t*)g.c_rowptr.data_ptr(), cur_stream()));
  } else {
    csr_by_key(center, n_nodes, g.c_rowptr, g.c_perm);
  }
  const char* env = std::getenv("XEQ_PBC_MIRROR");
  const bool periodic = cell_offsets != nullptr && cell_offsets->defined();
  g.mirror = symmetric && center_sorted && (!periodic || !(env && env[0] == '0' && env[1] == 0));
  if (g.mirror && !periodic) {
    g.n_rowptr = g.c_rowptr;
    g.n_perm = i32(g.E, g.ei);
    XCALL(xeq_reverse_edge_map((const int64_t*)g.ei.data_ptr(), g.E, n_nodes, (const int32_t*)g.c_rowptr.data_ptr(),
                               (int32_t*)g.n_perm.data_ptr(), cur_stream()));
    g.mirror_map = g.n_perm;
  } else if (g.mirror) {
    g.mirror_map = i32(g.E, g.ei);
    XCALL(xeq_reverse_edge_map_pbc(dcode(*cell_offsets), (const int64_t*)g.ei.data_ptr(), cell_offsets->data_ptr(), g.E, n_nodes,
                                   (const int32_t*)g.c_rowptr.data_ptr(), (int32_t*)g.mirror_map.data_ptr(), cur_stream()));
  } else {
    g.sorted_view();
  }
  return g;
}

void build_wq_plan(Graph& g, bool reverse, WqPlan& p) {
  if (reverse) g.sorted_view();
  const int eps = xeq_message_wq_edges_per_stream(g.N, g.E);   // (the C ABI states the rule; ops._wq_edges_per_stream asks it too)
  p.n_ranges = (int)std::max<int64_t>(1, (g.E + 2 * eps - 1) / (2 * eps));
  p.pcap = xeq_message_wq_pcap(g.N, g.E);
  p.qptr = i32(g.N + 1, g.ei);
  p.pgath = i32(p.pcap, g.ei);
  p.peid = i32(p.pcap, g.ei);
  p.qinfo = i32(p.pcap / 4, g.ei);
  p.sq = i32(2 * p.n_ranges + 1, g.ei);
  p.sn = i32(2 * p.n_ranges + 1, g.ei);
  p.win = i32(xeq_message_wq_win_ints(p.n_ranges), g.ei);
  const int64_t wbytes = xeq_message_wq_plan_workspace(g.N);
  TORCH_CHECK(wbytes >= 0, "xequinet_amd: wq plan workspace");
  p.work = at::empty({std::max<int64_t>(wbytes, 1)}, g.ei.options().dtype(at::kByte));
  p.rowptr = reverse ? g.n_rowptr : g.c_rowptr;
  const Tensor& perm = reverse ? g.n_perm : g.c_perm;
  const Tensor owner = g.ei.select(0, reverse ? 1 : 0), gather = g.ei.select(0, reverse ? 0 : 1);
  XCALL(xeq_message_wq_plan((const int32_t*)p.rowptr.data_ptr(), perm.defined() ? (const int32_t*)perm.data_ptr() : nullptr,
                            (const int64_t*)owner.data_ptr(), (const int64_t*)gather.data_ptr(), g.N, g.E, p.n_ranges,
                            p.work.data_ptr(), wbytes, (int32_t*)p.qptr.data_ptr(), (int32_t*)p.pgath.data_ptr(),
                            (int32_t*)p.peid.data_ptr(), (int32_t*)p.qinfo.data_ptr(), (int32_t*)p.sq.data_ptr(),
                            (int32_t*)p.sn.data_ptr(), (int32_t*)p.win.data_ptr(), cur_stream()));
}

// ---------------------------------------------------------------------------------------------- model description
// iparams: [node_dim, mul0, mul1, mul2, num_basis, n_blocks, rbf_kind, cutoff_kind, layer_norm, embed_kind, xhat_unused]
// fparams: [cutoff, invariant_eps]
// params (flat, in this order; undefined-by-absence entries are passed as empty tensors):
//   0 embed_table [87, A] (embed_kind 0) or embedding matrix [100, F] (embed_kind 1);  1 embed_w [F, A];  2 embed_b [F]
//   3 rbf_p0 [B];  4 rbf_p1 [B] (gaussian std; empty for bessel)
//   per block i (MSG = 5 + 27 i):
//     +0 mlp0_w [F,F]  +1 mlp0_b  +2 mlp2_w [H,F]  +3 mlp2_b  +4 rbf_w [H,B]  +5 rbf_b  +6 ln_w  +7 ln_b  +8 eq_w [C]  +9 eq_b [F]
//     +10 uv_pack_l0 [mul0, 2 mul0]  +11 uv_pack_l1  +12 uv_pack_l2  +13 uv_bias [2 F]
//     +14 dot_w [F, C]  +15 mlp3_w [F, F+C]  +16 mlp3_b  +17 mlp4_w [C+2F, F]  +18 mlp4_b  +19 ln_w  +20 ln_b  +21 eq_w  +22 eq_b
//     (+23..26 reserved)
//   tail: out0_w [Hd, F], out0_b, out2_w [1, Hd], out2_b
constexpr int P_BLOCK0 = 5, P_PER_BLOCK = 27;
struct Hyper {
  int F, mul[3], B, blocks, rbf_kind, cutoff_kind, layer_norm, embed_kind;
  double cutoff, inv_eps;
  int C() const { return mul[0] + mul[1] + mul[2]; }
  int D() const { return mul[0] + 3 * mul[1] + 5 * mul[2]; }
  int H() const { return F + 2 * C(); }
};

const Tensor* opt(const Tensor& t) { return t.defined() && t.numel() > 0 ? &t : nullptr; }
const void* OP(const Tensor& t) { return t.defined() && t.numel() > 0 ? t.data_ptr() : nullptr; }

// views of a BT buffer as plain matrices [n (2l+1), width * mul_l] per non-empty l  (nn/fused.py::_bt_blocks)
struct BtBlock {
  int l, m;
  Tensor view;
};
std::vector<BtBlock> bt_blocks(const Tensor& buf, int64_t n, const int mul[3], int width) {
  std::vector<BtBlock> out;
  int64_t base = 0;
  for (int l = 0; l < 3; ++l) {
    const int d = 2 * l + 1, m = mul[l];
    if (m > 0) out.push_back({l, m, buf.narrow(0, n * base * width, n * d * m * width).view({n * d, (int64_t)width * m})});
    base += (int64_t)d * m;
  }
  return out;
}

struct NormOut {
  Tensor shat, xhat, stats;
};
NormOut norm_fwd(const Hyper& hy, const Tensor& s, const Tensor& x, const Tensor& lw, const Tensor& lb, const Tensor& ew,
                 const Tensor& eb, Tensor shat_out, int64_t ld) {
  const int64_t n = x.size(0);
  NormOut o;
  if (!shat_out.defined()) {
    shat_out = at::empty({n, hy.F}, s.options());
    ld = hy.F;
  }
  o.shat = shat_out;
  o.xhat = at::empty({n * hy.D()}, x.options());
  o.stats = at::empty({n, 4}, s.options());
  const int32_t mul[3] = {hy.mul[0], hy.mul[1], hy.mul[2]};
  XCALL(xeq_norm_fwd(dcode(s), s.data_ptr(), x.data_ptr(), hy.layer_norm ? lw.data_ptr() : nullptr,
                     hy.layer_norm ? lb.data_ptr() : nullptr, hy.layer_norm ? ew.data_ptr() : nullptr,
                     hy.layer_norm ? eb.data_ptr() : nullptr, n, hy.F, mul, hy.layer_norm, o.shat.data_ptr(), ld,
                     o.xhat.data_ptr(), o.stats.data_ptr(), cur_stream()));
  return o;
}
// Front half of the FIRST message block from the element table (nn/fused.py::first_block_front is the Python twin): behind the embedding a
// node's scalars are a function of its element and x = 0, so LayerNorm, EquivariantLayerNorm and scalar_mlp (nn/xpainn.py:128-139) are
// evaluated once per table row -- the same xeq_linear_fwd / xeq_norm_fwd / xeq_mlp2_fwd launches, which give a row the same bits in any
// batch -- and cached per weight version; an evaluation gathers (s, h, xhat's 0e block) by atomic number in ONE launch
// (xeq_first_block_front).  q: the first block's parameters.
struct ElementFront {
  std::vector<int64_t> key;
  Owners owners;
  Tensor rows_s, rows_h, rows_x0;
};
const ElementFront* element_front(const Hyper& hy, const Tensor& table, const Tensor& ew, const Tensor& eb, const Tensor* q) {
  static std::mutex mu;
  static std::unordered_map<const void*, ElementFront> cache;
  const Tensor* ts[11] = {&table, &ew, &eb, &q[0], &q[1], &q[2], &q[3], &q[6], &q[7], &q[8], &q[9]};
  std::vector<int64_t> key;
  key.push_back((int64_t)xeq_pack_epoch());
  for (const Tensor* t : ts) {
    key.push_back((int64_t)t->_version());
    key.push_back((int64_t)(intptr_t)t->data_ptr());
  }
  std::lock_guard<std::mutex> lock(mu);
  ElementFront& e = cache[table.data_ptr()];
  const bool same = e.rows_s.defined() && e.key == key &&
                    e.owners.same({ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], ts[7], ts[8], ts[9], ts[10]});
  if (!same) {
    const int64_t zt = table.size(0);
    const Tensor z_all = at::arange(zt, table.options().dtype(at::kInt));
    e.rows_s = linear_fwd(table, ew, eb, 0, &z_all, nullptr);
    const Tensor x0 = at::zeros({zt, (int64_t)hy.D()}, table.options());
    NormOut no = norm_fwd(hy, e.rows_s, x0, q[6], q[7], q[8], q[9], Tensor(), 0);
    Tensor pre;
    mlp_fwd(no.shat, q[0], q[1], q[2], q[3], pre, e.rows_h);
    e.rows_h = e.rows_h.contiguous();
    e.rows_x0 = no.xhat.slice(0, 0, zt * hy.F).view({zt, (int64_t)hy.F}).contiguous();   // BT layout: the 0e block comes first
    e.owners.set({ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], ts[7], ts[8], ts[9], ts[10]});
    e.key = key;
  }
  return &e;
}
void norm_bwd(const Hyper& hy, const Tensor& s, const Tensor& x, const Tensor& lw, const Tensor& ew, const Tensor& stats,
              const Tensor& g_shat, int64_t ld, const Tensor& g_xhat, const Tensor& res_s, const Tensor& res_x, Tensor& g_s,
              Tensor& g_x) {
  const int64_t n = x.size(0);
  g_s = at::empty_like(s);
  g_x = at::empty_like(x);
  const int32_t mul[3] = {hy.mul[0], hy.mul[1], hy.mul[2]};
  XCALL(xeq_norm_bwd(dcode(s), s.data_ptr(), x.data_ptr(), hy.layer_norm ? lw.data_ptr() : nullptr,
                     hy.layer_norm ? ew.data_ptr() : nullptr, stats.data_ptr(), n, hy.F, mul, hy.layer_norm, g_shat.data_ptr(), ld,
                     g_xhat.data_ptr(), res_s.defined() ? res_s.data_ptr() : nullptr, res_x.defined() ? res_x.data_ptr() : nullptr,
                     g_s.data_ptr(), g_x.data_ptr(), cur_stream()));
}

// what one block keeps for the reverse pass
// The seed of an inference reverse pass behind the fused head: MINUS one (a [1] f32 constant, cached per device; nn/basic.py::_seed and
// nn/fused.py::constant_vector are the Python twins).  The reverse pass then returns -dE/dx = the forces without a negation launch -- and
// with the bits of the Python front: the bf16 matrix instructions are not symmetric in the sign (profiles/r05_mfma_sign.txt), so a pass
// seeded with +1 and negated afterwards differs in the last bit wherever a cotangent runs through the fused node block.  Not cached while
// a HIP graph is being captured (the tensor would live in that graph's pool).
Tensor minus_one(const at::TensorOptions& fopt) {
  static std::mutex mu;
  static std::unordered_map<int, Tensor> cache;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing((hipStream_t)cur_stream(), &capturing);
  std::lock_guard<std::mutex> lock(mu);
  const int dev = (int)fopt.device().index();
  auto it = cache.find(dev);
  if (it != cache.end()) return it->second;
  Tensor t = at::full({1}, -1.0, fopt.dtype(at::kFloat));
  if (capturing == hipStreamCaptureStatusNone) {
    (void)hipStreamSynchronize((hipStream_t)cur_stream());   // once: complete before any other stream may read it
    cache[dev] = t;
  }
  return t;
}

struct MsgSaved {
  Tensor s, x, stats, pre, h, xhat;
  int impl = 0;   // 0 wq, 1 sb
};
struct UpdSaved {
  Tensor s, x, stats, uv, pre, a, ip;
};

// ---------------------------------------------------------------------------------------------- the evaluation
std::vector<Tensor> xpainn_eval_impl(const Tensor& pos_in, const Tensor& atomic_numbers, const Tensor& edge_index, const Tensor& ptr,
                                     const c10::optional<Tensor>& cell_o, const c10::optional<Tensor>& cell_offsets_o,
                                     const std::vector<Tensor>& prm, const std::vector<int64_t>& ip, const std::vector<double>& fp,
                                     bool center_sorted, bool symmetric, bool compute_forces, bool compute_virial) {
  need_hip(pos_in, "pos");
  need_hip(edge_index, "edge_index");
  TORCH_CHECK(ip.size() >= 10 && fp.size() >= 2, "xeq::xpainn_eval: malformed hyper-parameter lists");
  Hyper hy;
  hy.F = (int)ip[0];
  hy.mul[0] = (int)ip[1];
  hy.mul[1] = (int)ip[2];
  hy.mul[2] = (int)ip[3];
  hy.B = (int)ip[4];
  hy.blocks = (int)ip[5];
  hy.rbf_kind = (int)ip[6];
  hy.cutoff_kind = (int)ip[7];
  hy.layer_norm = (int)ip[8];
  hy.embed_kind = (int)ip[9];
  hy.cutoff = fp[0];
  hy.inv_eps = fp[1];
  TORCH_CHECK((int64_t)prm.size() == P_BLOCK0 + (int64_t)P_PER_BLOCK * hy.blocks + 4, "xeq::xpainn_eval: expected ",
              P_BLOCK0 + P_PER_BLOCK * hy.blocks + 4, " parameter tensors, got ", prm.size());
  TORCH_CHECK(edge_index.dim() == 2 && edge_index.size(0) == 2 && edge_index.scalar_type() == at::kLong, "edge_index must be int64 [2, E]");
  const Tensor pos = pos_in.detach().contiguous();
  const auto fopt = pos.options();
  const int dt = dcode(pos);
  const int64_t N = pos.size(0), G = ptr.numel() - 1;
  const int C = hy.C(), D = hy.D(), H = hy.H(), F = hy.F;
  const int32_t mul[3] = {hy.mul[0], hy.mul[1], hy.mul[2]};
  const Tensor ptr64 = ptr.to(at::kLong).contiguous();
  void* st = cur_stream();

  const Tensor ei_c = edge_index.contiguous();
  const int64_t E = ei_c.size(1);

  // ---- sorted views of the edge list (ops.EdgeGraph), then the edge geometry (nn/basic.py:110-131): the order of the Python front
  // (nn/basic.py::compute_edge_data builds the graph first), so that both fronts issue ONE launch sequence
  // (tests/test_gpu_interface.py::test_both_fronts_issue_the_same_launch_sequence)
  Tensor cell, cell_offsets, batch;
  const bool has_cell = cell_o.has_value() && cell_o->defined();
  if (has_cell) {
    TORCH_CHECK(cell_offsets_o.has_value() && cell_offsets_o->defined(), "cell without cell_offsets");
    cell = cell_o->to(pos.scalar_type()).contiguous();
    cell_offsets = cell_offsets_o->to(pos.scalar_type()).contiguous();
  }
  Graph g = build_graph(ei_c, N, center_sorted, symmetric, has_cell ? &cell_offsets : nullptr);
  if (has_cell) {
    if (G > 1) {
      const Tensor counts = ptr64.slice(0, 1) - ptr64.slice(0, 0, G);
      batch = at::repeat_interleave(at::arange(G, ptr64.options()), counts, 0, N);
    }
  }
  Tensor vec = at::empty({E, 3}, fopt), dist = at::empty({E}, fopt);
  XCALL(xeq_edge_vectors_fwd(dt, pos.data_ptr(), (const int64_t*)ei_c.data_ptr(), E, has_cell ? cell.data_ptr() : nullptr,
                             has_cell ? cell_offsets.data_ptr() : nullptr, batch.defined() ? (const int64_t*)batch.data_ptr() : nullptr,
                             vec.data_ptr(), dist.data_ptr(), st));

  // ---- which message kernels (ops.select_message_impl, without the generic form).  The family rule is the C ABI's
  // (xeq_message_auto_family: the Python modules ask the same function); this operator carries the wq and sb sequences
  int impl;
  const int family = xeq_message_auto_family(dt, N, E, hy.B, F, mul);
  if (family == XEQ_FAMILY_WQ) impl = 0;
  else if (family == XEQ_FAMILY_SB) impl = 1;
  else TORCH_CHECK(false, "xeq::xpainn_eval: this configuration / size needs the generic message kernels: use the Python modules");

  // ---- embedding (nn/xpainn.py:55-83) -- and, where the table form covers the layout, the first block's norms and scalar_mlp with it:
  // one gather by atomic number from per-element rows (element_front above; nn/fused.py::first_block_front)
  std::vector<MsgSaved> msv(hy.blocks);
  std::vector<UpdSaved> usv(hy.blocks);
  Tensor s;
  Tensor x = at::zeros({N, D}, fopt);
  bool front_done = false;
  const bool z_int = atomic_numbers.scalar_type() == at::kInt || atomic_numbers.scalar_type() == at::kLong;
  if (hy.embed_kind == 0 && hy.blocks > 0 && dt == XEQ_F32 && hy.layer_norm && hy.mul[0] == F && prm[0].scalar_type() == at::kFloat &&
      prm[0].dim() == 2 && prm[0].stride(1) == 1 && prm[0].stride(0) % 4 == 0 && lin_pack(prm[1], prm[2]) != nullptr) {
    const ElementFront* ef = element_front(hy, prm[0], prm[1], prm[2], &prm[P_BLOCK0]);
    const Tensor z = (z_int ? atomic_numbers : atomic_numbers.to(at::kLong)).contiguous();
    MsgSaved& m = msv[0];
    s = at::empty({N, (int64_t)F}, fopt);
    m.h = at::empty({N, (int64_t)H}, fopt);
    m.xhat = at::empty({N * (int64_t)D}, fopt);
    // the wq kernels never read xhat's l > 0 blocks behind the embedding (XEQ_XHAT_HIGHER_L_ZERO): those are then not even written
    XCALL(xeq_first_block_front(z.data_ptr(), z.scalar_type() == at::kLong, N, ef->rows_s.size(0), ef->rows_s.data_ptr(),
                                ef->rows_h.data_ptr(), ef->rows_x0.data_ptr(), F, H, impl == 0 ? F : D, s.data_ptr(), m.h.data_ptr(),
                                m.xhat.data_ptr(), st));
    m.s = s;
    m.x = x;
    front_done = true;
  } else if (hy.embed_kind == 0) {
    const Tensor z32 = atomic_numbers.to(at::kInt).contiguous();
    s = linear_fwd(prm[0], prm[1], prm[2], 0, &z32);   // table lookup + Linear in one launch (nn/xpainn.py::XEmbedding._embed)
  }
  else s = prm[0].index_select(0, atomic_numbers.to(at::kLong));
  const Tensor& p0 = prm[3];
  const Tensor& p1 = prm[4];

  // ---- the first block's norms and scalar MLP where the table form above does not cover the layout (per-node launches)
  if (hy.blocks > 0 && !front_done) {
    const Tensor* q = &prm[P_BLOCK0];
    MsgSaved& m = msv[0];
    m.s = s;
    m.x = x;
    NormOut no = norm_fwd(hy, s, x, q[6], q[7], q[8], q[9], Tensor(), 0);
    m.stats = no.stats;
    m.xhat = no.xhat;
    mlp_fwd(no.shat, q[0], q[1], q[2], q[3], m.pre, m.h);
  }
  if (impl == 0) {
    build_wq_plan(g, false, g.fwd);
    g.fwd.basis = at::empty({g.fwd.pcap, xeq_message_wq_record_floats_for(hy.B)}, fopt);
    // a reverse pass over the same plan (mirror walk) follows: its derivative records come out of the same launch (ops.message_forward)
    if (g.mirror && (compute_forces || compute_virial)) g.fwd.dbasis = at::empty({g.fwd.pcap, xeq_message_wq_record_floats_for(hy.B)}, fopt);
    XCALL(xeq_edge_basis_wq(vec.data_ptr(), N, E, (const int32_t*)g.fwd.qptr.data_ptr(), (const int32_t*)g.fwd.peid.data_ptr(),
                            hy.rbf_kind, hy.cutoff_kind, hy.B, hy.cutoff, p0.data_ptr(), OP(p1), g.fwd.basis.data_ptr(),
                            g.fwd.dbasis.defined() ? g.fwd.dbasis.data_ptr() : nullptr, st));
  } else {
    const int w = xeq_edge_basis_width(hy.B);
    g.sb_basis = at::empty({E, w}, fopt);
    g.sb_dbasis = at::empty({E, w}, fopt);
    XCALL(xeq_edge_basis(dt, vec.data_ptr(), E, hy.rbf_kind, hy.cutoff_kind, hy.B, hy.cutoff, p0.data_ptr(), OP(p1),
                         g.sb_basis.data_ptr(), g.sb_dbasis.data_ptr(), st));
  }

  // the wq kernels take rbf_lin's rows from the packed copy (one per weight version)
  auto wq_w = [&](const Tensor* q) {
    const Tensor* pk = wq_weight_pack(q[4], q[5], hy.B, F, mul);
    TORCH_CHECK(pk != nullptr, "xeq::xpainn_eval: rbf_lin weights cannot be packed for the wq kernels");
    return pk;
  };
  // fused node blocks: f32, the default layout, layer norms on (nn/nodeblock.py::supported)
  const bool nb_ok = dt == XEQ_F32 && hy.layer_norm && xeq_node_block_supported(XEQ_F32, F, mul) && xeq_node_block_auto(N);
  for (int b = 0; b < hy.blocks; ++b) {
    const Tensor* q = &prm[P_BLOCK0 + P_PER_BLOCK * b];
    {  // ---- XPainnMessage.forward (nn/xpainn.py:128-161; nn/fused.py::MessageBlock)
      MsgSaved& m = msv[b];
      if (b > 0 && !m.h.defined()) {   // (block 0: done above; behind a fused node block: done by its launch)
        m.s = s;
        m.x = x;
        NormOut no = norm_fwd(hy, s, x, q[6], q[7], q[8], q[9], Tensor(), 0);
        m.stats = no.stats;
        m.xhat = no.xhat;
        mlp_fwd(no.shat, q[0], q[1], q[2], q[3], m.pre, m.h);
      }
      Tensor s_out = at::empty_like(s), x_out = at::empty_like(x);
      m.impl = impl;
      if (impl == 0) {
        XCALL(xeq_message_fwd_wq(N, E, g.fwd.n_ranges, (const int32_t*)g.fwd.sq.data_ptr(), (const int32_t*)g.fwd.sn.data_ptr(),
                                 (const int32_t*)g.fwd.win.data_ptr(), (const int32_t*)g.fwd.rowptr.data_ptr(),
                                 (const int32_t*)g.fwd.pgath.data_ptr(), (const int32_t*)g.fwd.qinfo.data_ptr(), g.fwd.basis.data_ptr(),
                                 m.h.data_ptr(), m.xhat.data_ptr(), s.data_ptr(), x.data_ptr(), wq_w(q)->data_ptr(), nullptr, hy.B,
                                 F, mul, s_out.data_ptr(), x_out.data_ptr(), (b == 0 ? (1 | XEQ_XHAT_HIGHER_L_ZERO) : 1) | XEQ_WQ_PACKED_WEIGHTS, st));   // block 0: x = 0
      } else {
        XCALL(xeq_message_fwd_sb(dt, N, E, (const int32_t*)g.c_rowptr.data_ptr(),
                                 g.c_perm.defined() ? (const int32_t*)g.c_perm.data_ptr() : nullptr,
                                 (const int64_t*)g.ei.select(0, 1).data_ptr(), g.sb_basis.data_ptr(), m.h.data_ptr(), m.xhat.data_ptr(),
                                 s.data_ptr(), x.data_ptr(), q[4].data_ptr(), q[5].data_ptr(), hy.B, F, mul, s_out.data_ptr(),
                                 x_out.data_ptr(), 1, st));
      }
      s = s_out;
      x = x_out;
    }
    {  // ---- XPainnUpdate.forward (nn/xpainn.py:206-231; nn/fused.py::UpdateBlock)
      UpdSaved& u = usv[b];
      u.s = s;
      u.x = x;
      const bool last_blk = b == hy.blocks - 1;
      if (nb_ok) {   // the whole block, and the front half of the next message block, in one launch (nn/fused.py::NodeBlock)
        const Tensor* qn = last_blk ? nullptr : &prm[P_BLOCK0 + P_PER_BLOCK * (b + 1)];
        const NbPacks* pk = nb_packs(q, qn, !last_blk);
        auto fp = [](const Tensor& t) { return t.defined() ? (const float*)t.data_ptr() : nullptr; };
        auto fpm = [](Tensor& t) { return t.defined() ? (float*)t.data_ptr() : nullptr; };
        const int64_t NR = xeq_node_block_rows(N);   // internal tensors: whole workgroups, wave-native layout
        u.uv = at::empty({2 * NR * D}, fopt);
        u.stats = at::empty({N, 4}, fopt);
        u.pre = at::empty({NR, F}, fopt);
        u.a = at::empty({NR, C + 2 * F}, fopt);
        u.ip = at::empty({NR, F}, fopt);
        Tensor p_scr = at::empty({NR, C}, fopt);
        Tensor s_out = at::empty_like(s), x_out = last_blk ? Tensor() : at::empty_like(x);
        MsgSaved* mn = last_blk ? nullptr : &msv[b + 1];
        if (mn) {
          mn->stats = at::empty({N, 4}, fopt);
          mn->xhat = at::empty({N * D}, fopt);
          mn->pre = at::empty({NR, F}, fopt);
          mn->h = at::empty({N, H}, fopt);
        }
        XCALL(xeq_node_block_fwd(N, fp(s), fp(x), fp(q[19]), fp(q[20]), fp(q[21]), fp(q[22]), fp(pk->bias_uv), fp(q[16]), fp(q[18]),
                                 hy.inv_eps, pk->fwd.data_ptr(), fpm(p_scr), fpm(u.uv), fpm(u.stats), fpm(u.pre), fpm(u.a), fpm(u.ip),
                                 fpm(s_out), fpm(x_out), qn ? fp(qn[6]) : nullptr, qn ? fp(qn[7]) : nullptr, qn ? fp(qn[8]) : nullptr,
                                 qn ? fp(qn[9]) : nullptr, qn ? fp(qn[1]) : nullptr, qn ? fp(qn[3]) : nullptr,
                                 mn ? fpm(mn->stats) : nullptr, mn ? fpm(mn->xhat) : nullptr, mn ? fpm(mn->pre) : nullptr,
                                 mn ? fpm(mn->h) : nullptr, st));
        s = s_out;
        x = x_out;
        if (mn) {
          mn->s = s;
          mn->x = x;
        }
        continue;
      }
      Tensor cat = at::empty({N, F + C}, fopt);
      u.uv = at::empty({2 * N * D}, fopt);
      Tensor p = at::empty({N, C}, fopt);
      if (const UvFrag* fr = uv_frag(&q[10], F, mul)) {   // norms -> U, V -> v, p in one matrix-core launch
        u.stats = at::empty({N, 4}, fopt);
        auto fp = [](const Tensor& t) { return t.defined() ? (const float*)t.data_ptr() : nullptr; };
        XCALL(xeq_update_uv_fwd((const float*)s.data_ptr(), (const float*)x.data_ptr(), hy.layer_norm ? fp(q[19]) : nullptr,
                                hy.layer_norm ? fp(q[20]) : nullptr, hy.layer_norm ? fp(q[21]) : nullptr,
                                hy.layer_norm ? fp(q[22]) : nullptr, N, F, mul, hy.layer_norm, fp(fr->w[0]), fp(fr->w[1]), fp(fr->w[2]),
                                q[13].numel() > 0, hy.inv_eps, (float*)cat.data_ptr(), F + C, (float*)p.data_ptr(),
                                (float*)u.uv.data_ptr(), (float*)u.stats.data_ptr(), st));
      } else {
        NormOut no = norm_fwd(hy, s, x, q[19], q[20], q[21], q[22], cat, F + C);
        u.stats = no.stats;
        auto xb = bt_blocks(no.xhat, N, hy.mul, 1), ub = bt_blocks(u.uv, N, hy.mul, 2);
        for (size_t k = 0; k < xb.size(); ++k) {
          const Tensor& W = q[10 + xb[k].l];
          if (xb[k].l == 0 && q[13].numel() > 0) at::addmm_out(ub[k].view, q[13], xb[k].view, W);
          else at::mm_out(ub[k].view, xb[k].view, W);
        }
        XCALL(xeq_uv_reduce_fwd(dt, u.uv.data_ptr(), N, mul, hy.inv_eps, cat.data_ptr(), F + C, F, p.data_ptr(), st));
      }
      mlp_and_linear_fwd(cat, q[15], q[16], q[17], q[18], p, q[14], u.pre, u.a, u.ip);
      const bool last = b == hy.blocks - 1;   // the energy head reads the scalars only: the last equivariant output has no consumer
      Tensor s_out = at::empty_like(s), x_out = last ? Tensor() : at::empty_like(x);
      XCALL(xeq_update_out_fwd(dt, s.data_ptr(), x.data_ptr(), u.uv.data_ptr(), u.a.data_ptr(), u.ip.data_ptr(), N, F, mul,
                               s_out.data_ptr(), last ? nullptr : x_out.data_ptr(), st));
      s = s_out;
      x = x_out;
    }
  }
  // ---- EnergyOut.forward (nn/output.py:114-128)
  const Tensor* t = &prm[P_BLOCK0 + P_PER_BLOCK * hy.blocks];
  // (nn/fused.py::EnergyHead is the Python twin: the same kernels where they take the layer, the library GEMMs elsewhere)
  const LinPack* head_pk = dt == XEQ_F32 && t[0].size(0) % 4 == 0 && t[2].size(0) == 1 ? lin_pack(t[0], t[1]) : nullptr;
  const bool head_native = head_pk != nullptr && head_pk->bwd.defined();
  Tensor pre_o, atomic, jac;
  // round 5: the head with its whole reverse pass saved as one row per node, in ONE launch (nn/fused.py::EnergyReadout is the twin)
  const bool head_fused = head_native && s.stride(1) == 1 && s.stride(0) % 4 == 0 && t[1].defined() && t[1].numel() > 0 &&
                          xeq_head_supported(XEQ_F32, F, (int)t[0].size(0));
  if (head_fused) {
    atomic = at::empty({N}, fopt);
    if (compute_forces || compute_virial) jac = at::empty({N, (int64_t)F}, fopt);
    XCALL(xeq_head_fwd(s.data_ptr(), s.stride(0), N, F, (int)t[0].size(0), head_pk->fwd.data_ptr(), head_pk->bwd.data_ptr(), t[2].data_ptr(),
                       t[3].data_ptr(), atomic.data_ptr(), jac.defined() ? jac.data_ptr() : nullptr, st));
  } else if (head_native) {
    const Tensor hidden = linear_fwd(s, t[0], t[1], 1, nullptr, &pre_o);
    atomic = at::empty({N}, fopt);
    XCALL(xeq_head_dot(hidden.data_ptr(), N, (int)t[0].size(0), t[2].data_ptr(), t[3].data_ptr(), atomic.data_ptr(), st));
  } else {
    pre_o = at::addmm(t[1], s, t[0].t());
    atomic = at::addmm(t[3], at::silu(pre_o), t[2].t()).reshape({-1});
  }
  Tensor energy = at::empty({G}, fopt);
  XCALL(xeq_segment_sum(dt, atomic.data_ptr(), (const int64_t*)ptr64.data_ptr(), G, 1, energy.data_ptr(), st));

  Tensor forces, virial;
  if (compute_forces || compute_virial) {
    // ---- explicit reverse pass: dE/ds of the head, then the blocks backwards, then the edge geometry (nn/basic.py:143-199)
    Tensor g_s;   // dE_i/d atomic_i = 1
    const bool neg_seed = head_fused;   // the reverse pass runs on MINUS the gradient: forces (and the virial) come out without a negation
    if (head_fused) {
      // (seed) x d atomic_i / d s_i, the row saved by the forward launch (nn/fused.py::EnergyReadout.backward)
      const Tensor seed = minus_one(fopt);
      g_s = at::empty_like(jac);
      XCALL(xeq_head_bwd(jac.data_ptr(), N, F, nullptr, 0, seed.data_ptr(), 0, nullptr, g_s.data_ptr(), st));
    } else if (head_native) {
      Tensor g_hidden = at::empty_like(pre_o);
      XCALL(xeq_head_bwd_hidden(pre_o.data_ptr(), N, (int)pre_o.size(1), t[2].data_ptr(), nullptr, g_hidden.data_ptr(), st));
      g_s = linear_bwd(g_hidden, t[0], t[1]);
    } else {
      g_s = at::mm(at::silu_backward(t[2].expand({N, t[2].size(1)}), pre_o), t[0]);
    }
    Tensor g_x;   // undefined = zero: the head reads the scalars only, the last block's equivariant output has no consumer
    Tensor g_vec_total;
    // round 5: the wq blocks' partials are added up and the chain rule to dL/dvec runs ONCE behind the last block of the reverse pass
    // (xeq_message_wq_edge_grad_sum; ops.EdgeGradDeferral is the Python twin, same order: last block first)
    const bool defer_edge_grad = impl == 0 && hy.blocks <= XEQ_WQ_MAX_PART_SETS;
    std::vector<Tensor> part_sets;
    if (impl == 0 && !g.mirror) {
      build_wq_plan(g, true, g.rev);
      g.rev.basis = at::empty({g.rev.pcap, xeq_message_wq_record_floats_for(hy.B)}, fopt);
      g.rev.dbasis = at::empty({g.rev.pcap, xeq_message_wq_record_floats_for(hy.B)}, fopt);
      XCALL(xeq_edge_basis_wq(vec.data_ptr(), N, E, (const int32_t*)g.rev.qptr.data_ptr(), (const int32_t*)g.rev.peid.data_ptr(),
                              hy.rbf_kind, hy.cutoff_kind, hy.B, hy.cutoff, p0.data_ptr(), OP(p1), g.rev.basis.data_ptr(),
                              g.rev.dbasis.data_ptr(), st));
    }
    Tensor pend_gh, pend_gxhat;   // gradients of the next block's (h, xhat): reversed by the node block of the update in front of it
    for (int b = hy.blocks - 1; b >= 0; --b) {
      const Tensor* q = &prm[P_BLOCK0 + P_PER_BLOCK * b];
      if (nb_ok) {  // NodeBlock.backward
        const UpdSaved& u = usv[b];
        const bool last_blk = b == hy.blocks - 1;
        const Tensor* qn = last_blk ? nullptr : &prm[P_BLOCK0 + P_PER_BLOCK * (b + 1)];
        const NbPacks* pk = nb_packs(q, qn, !last_blk);
        const MsgSaved* mn = last_blk ? nullptr : &msv[b + 1];
        auto fp = [](const Tensor& t) { return t.defined() ? (const float*)t.data_ptr() : nullptr; };
        Tensor ns = at::empty_like(u.s), nx = at::empty_like(u.x);
        const int64_t NR = xeq_node_block_rows(N);
        Tensor gxo = last_blk ? Tensor() : at::empty({NR, D}, fopt);
        Tensor gp = at::empty({NR, C}, fopt), gv = at::empty({NR, C}, fopt), gw = at::empty({NR, D}, fopt);
        if (!last_blk && !g_x.defined()) g_x = at::zeros({N, D}, fopt);
        XCALL(xeq_node_block_bwd(N, fp(pend_gh), fp(pend_gxhat), fp(g_s), fp(g_x), mn ? fp(mn->s) : nullptr, mn ? fp(mn->x) : nullptr,
                                 mn ? fp(mn->stats) : nullptr, mn ? fp(mn->pre) : nullptr, qn ? fp(qn[6]) : nullptr,
                                 qn ? fp(qn[8]) : nullptr, fp(u.uv), fp(u.a), fp(u.ip), fp(u.pre), fp(u.s), fp(u.x), fp(u.stats), fp(q[19]),
                                 fp(q[21]), hy.inv_eps, pk->bwd.data_ptr(), gxo.defined() ? (float*)gxo.data_ptr() : nullptr,
                                 (float*)gp.data_ptr(), (float*)gv.data_ptr(), (float*)gw.data_ptr(), (float*)ns.data_ptr(),
                                 (float*)nx.data_ptr(), st));
        g_s = ns;
        g_x = nx;
        pend_gh = Tensor();
        pend_gxhat = Tensor();
      } else {  // UpdateBlock.backward
        const UpdSaved& u = usv[b];
        Tensor g_a = at::empty_like(u.a), g_ip = at::empty_like(u.ip);
        const void* gx_ptr = g_x.defined() ? g_x.data_ptr() : nullptr;
        XCALL(xeq_update_out_bwd(dt, g_s.data_ptr(), gx_ptr, u.uv.data_ptr(), u.a.data_ptr(), u.ip.data_ptr(), N, F, mul,
                                 g_a.data_ptr(), g_ip.data_ptr(), nullptr, st));
        Tensor g_p, g_cat;
        mlp_and_linear_bwd(g_a, u.pre, q[15], q[16], q[17], q[18], g_ip, q[14], g_cat, g_p);
        Tensor ns, nx;
        const UvFrag* fr = g_cat.is_contiguous() ? uv_frag(&q[10], F, mul) : nullptr;
        if (fr) {   // dL/dU, dL/dV -> dL/dxhat (-> reverse of both norms) in one matrix-core launch
          auto fp = [](const Tensor& t) { return t.defined() ? (const float*)t.data_ptr() : nullptr; };
          const bool fuse = N <= UV_BWD_FUSE_NORM_MAX_NODES;
          Tensor g_xhat;
          if (fuse) {
            ns = at::empty_like(u.s);
            nx = at::empty_like(u.x);
          } else {
            g_xhat = at::empty({N * D}, fopt);
          }
          XCALL(xeq_update_uv_bwd((const float*)u.uv.data_ptr(), (const float*)g_p.data_ptr(), (const float*)g_cat.data_ptr(), F + C,
                                  (const float*)gx_ptr, (const float*)g_s.data_ptr(), (const float*)u.a.data_ptr(), u.a.size(1),
                                  (const float*)u.s.data_ptr(), (const float*)u.x.data_ptr(), (const float*)u.stats.data_ptr(),
                                  hy.layer_norm ? fp(q[19]) : nullptr, hy.layer_norm ? fp(q[21]) : nullptr, N, F, mul, hy.layer_norm,
                                  fp(fr->wt[0]), fp(fr->wt[1]), fp(fr->wt[2]), hy.inv_eps, fuse ? (float*)ns.data_ptr() : nullptr,
                                  fuse ? (float*)nx.data_ptr() : nullptr, fuse ? nullptr : (float*)g_xhat.data_ptr(), st));
          if (!fuse) norm_bwd(hy, u.s, u.x, q[19], q[21], u.stats, g_cat, F + C, g_xhat, g_s, g_x, ns, nx);
        } else {
          Tensor g_uv = at::empty_like(u.uv);
          if (!g_x.defined()) g_x = at::zeros({N, D}, fopt);   // the kernel chain wants the tensor
          XCALL(xeq_uv_reduce_bwd(dt, u.uv.data_ptr(), g_p.data_ptr(), g_cat.data_ptr(), F + C, F, N, mul, hy.inv_eps, g_x.data_ptr(),
                                  u.a.data_ptr(), g_uv.data_ptr(), st));
          Tensor g_xhat = at::empty({N * D}, fopt);
          auto gb = bt_blocks(g_xhat, N, hy.mul, 1), gub = bt_blocks(g_uv, N, hy.mul, 2);
          for (size_t k = 0; k < gb.size(); ++k) at::mm_out(gb[k].view, gub[k].view, q[10 + gb[k].l].t());
          norm_bwd(hy, u.s, u.x, q[19], q[21], u.stats, g_cat, F + C, g_xhat, g_s, g_x, ns, nx);
        }
        g_s = ns;
        g_x = nx;
      }
      {  // MessageBlock.backward
        const MsgSaved& m = msv[b];
        const bool node_grads = b > 0 || m.impl != 0;   // first block: only dL/dvec leaves it, the wq kernel then stores no node gradients
        Tensor g_h = node_grads ? at::empty_like(m.h) : Tensor(), g_xhat = node_grads ? at::empty_like(m.xhat) : Tensor();
        Tensor g_vec = at::empty_like(vec);
        if (m.impl == 0) {
          Tensor parts = at::empty({std::max<int64_t>(1, xeq_message_wq_parts_floats(N, E, mul))}, fopt);
          const WqPlan& w = g.mirror ? g.fwd : g.rev;
          const int xl_bwd = (b == 0 ? (1 | XEQ_XHAT_HIGHER_L_ZERO) : 1) | (g.mirror ? XEQ_WQ_MIRROR_WALK : 0);
          XCALL(xeq_message_bwd_wq(N, E, w.n_ranges, (const int32_t*)w.sq.data_ptr(), (const int32_t*)w.sn.data_ptr(),
                                   (const int32_t*)w.win.data_ptr(), (const int32_t*)w.rowptr.data_ptr(),
                                   (const int32_t*)w.pgath.data_ptr(), (const int32_t*)w.qinfo.data_ptr(),
                                   w.basis.data_ptr(), w.dbasis.data_ptr(), m.h.data_ptr(), m.xhat.data_ptr(), g_s.data_ptr(),
                                   g_x.data_ptr(), wq_w(q)->data_ptr(), nullptr, hy.B, F, mul, node_grads ? g_h.data_ptr() : nullptr,
                                   node_grads ? g_xhat.data_ptr() : nullptr, parts.data_ptr(), xl_bwd | XEQ_WQ_PACKED_WEIGHTS, st));
          if (defer_edge_grad) {
            part_sets.push_back(parts);
            if (b == 0) {
              std::vector<const void*> pp;
              for (const Tensor& ps : part_sets) pp.push_back(ps.data_ptr());
              XCALL(xeq_message_wq_edge_grad_sum(vec.data_ptr(), N, E, (const int32_t*)w.qptr.data_ptr(), (const int32_t*)w.peid.data_ptr(),
                                                 g.mirror ? (const int32_t*)g.mirror_map.data_ptr() : nullptr, mul, (int)pp.size(), pp.data(),
                                                 g_vec.data_ptr(), st));
            }
          } else {
            XCALL(xeq_message_wq_edge_grad(vec.data_ptr(), N, E, (const int32_t*)w.qptr.data_ptr(), (const int32_t*)w.peid.data_ptr(),
                                           g.mirror ? (const int32_t*)g.mirror_map.data_ptr() : nullptr, mul, parts.data_ptr(),
                                           g_vec.data_ptr(), st));
          }
        } else {
          g.sorted_view();
          // the blocks share ONE dL/dvec buffer: the first to run stores, the others add (XEQ_SB_ACCUM_VEC; ops.message_backward does the same)
          const bool accum = g_vec_total.defined();
          if (!accum) g_vec_total = g_vec;
          XCALL(xeq_message_bwd_sb(dt, N, E, (const int32_t*)g.n_rowptr.data_ptr(), (const int32_t*)g.n_perm.data_ptr(),
                                   (const int64_t*)g.ei.select(0, 0).data_ptr(), g.sb_basis.data_ptr(), g.sb_dbasis.data_ptr(),
                                   m.h.data_ptr(), m.xhat.data_ptr(), g_s.data_ptr(), g_x.data_ptr(), q[4].data_ptr(), q[5].data_ptr(),
                                   hy.B, F, mul, g_h.data_ptr(), g_xhat.data_ptr(), g_vec_total.data_ptr(), 1 | (accum ? XEQ_SB_ACCUM_VEC : 0), st));
        }
        if (m.impl != 0) {
          // (summed in the kernel)
        } else if (defer_edge_grad) {
          if (b == 0) g_vec_total = g_vec;
        } else {
          g_vec_total = g_vec_total.defined() ? g_vec_total + g_vec : g_vec;
        }
        if (b == 0) break;   // the first block's node features (embedding, zeros) do not depend on the positions
        if (nb_ok) {         // the front half of this block is reversed by the node block of update b - 1; g_s, g_x: the residual path
          pend_gh = g_h;
          pend_gxhat = g_xhat;
          continue;
        }
        const Tensor g_shat = mlp_bwd(g_h, m.pre, q[0], q[1], q[2], q[3]);
        Tensor ns, nx;
        norm_bwd(hy, m.s, m.x, q[6], q[8], m.stats, g_shat, F, g_xhat, g_s, g_x, ns, nx);
        g_s = ns;
        g_x = nx;
      }
    }
    if (!g_vec_total.defined()) g_vec_total = at::zeros_like(vec);
    g_vec_total = g_vec_total.contiguous();
    if (compute_forces) {
      Tensor grad_pos = at::empty({N, 3}, fopt);
      XCALL(xeq_edge_vectors_bwd(dt, g_vec_total.data_ptr(), N, (const int32_t*)g.c_rowptr.data_ptr(),
                                 g.c_perm.defined() ? (const int32_t*)g.c_perm.data_ptr() : nullptr,
                                 (const int32_t*)g.rev_rowptr().data_ptr(), (const int32_t*)g.rev_perm().data_ptr(), grad_pos.data_ptr(), st));
      forces = neg_seed ? grad_pos : grad_pos.neg();
    }
    if (compute_virial) {   // sym(sum_e vec_e (x) dE/dvec_e) per graph, edges walked center-sorted (ops.EdgeVectors.backward)
      Tensor outer = (vec.unsqueeze(2) * g_vec_total.unsqueeze(1)).reshape({-1, 9});
      if (g.c_perm.defined()) outer = outer.index_select(0, g.c_perm.to(at::kLong));
      const Tensor eptr = g.c_rowptr.to(at::kLong).index_select(0, ptr64).contiguous();
      outer = outer.contiguous();
      Tensor msum = at::empty({G, 9}, fopt);
      XCALL(xeq_segment_sum(dt, outer.data_ptr(), (const int64_t*)eptr.data_ptr(), G, 9, msum.data_ptr(), st));
      msum = msum.view({G, 3, 3});
      virial = 0.5 * (msum + msum.transpose(1, 2));
      if (!neg_seed) virial = virial.neg();
    }
  }
  if (!forces.defined()) forces = at::empty({0, 3}, fopt);
  if (!virial.defined()) virial = at::empty({0, 3, 3}, fopt);
  return {energy, atomic, forces, virial};
}

// autograd wrapper: energy (and nothing else) is differentiable w.r.t. pos; backward = -forces * grad_energy[graph]
class XpainnEvalFn : public torch::autograd::Function<XpainnEvalFn> {
 public:
  static variable_list forward(AutogradContext* ctx, const Tensor& pos, const Tensor& atomic_numbers, const Tensor& edge_index,
                               const Tensor& ptr, const c10::optional<Tensor>& cell, const c10::optional<Tensor>& cell_offsets,
                               std::vector<Tensor> prm, std::vector<int64_t> ip, std::vector<double> fp, bool center_sorted,
                               bool symmetric, bool compute_forces, bool compute_virial) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const bool need = compute_forces || pos.requires_grad();
    auto out = xpainn_eval_impl(pos, atomic_numbers, edge_index, ptr, cell, cell_offsets, prm, ip, fp, center_sorted, symmetric, need,
                                compute_virial);
    ctx->save_for_backward({out[2], ptr});
    ctx->mark_non_differentiable({out[1], out[2], out[3]});
    return {out[0], out[1], out[2], out[3]};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor& forces = saved[0];
    const Tensor ptr = saved[1].to(at::kLong);
    Tensor g_pos;
    if (grads[0].defined() && forces.size(0) > 0) {
      const int64_t G = ptr.numel() - 1, N = forces.size(0);
      const Tensor counts = ptr.slice(0, 1) - ptr.slice(0, 0, G);
      const Tensor per_atom = at::repeat_interleave(grads[0].reshape({-1}), counts, 0, N);
      g_pos = forces.neg() * per_atom.unsqueeze(1);
    }
    variable_list r(13);
    r[0] = g_pos;
    return r;
  }
};

std::vector<Tensor> xpainn_eval(const Tensor& pos, const Tensor& atomic_numbers, const Tensor& edge_index, const Tensor& ptr,
                                const c10::optional<Tensor>& cell, const c10::optional<Tensor>& cell_offsets, std::vector<Tensor> prm,
                                std::vector<int64_t> ip, std::vector<double> fp, bool center_sorted, bool symmetric,
                                bool compute_forces, bool compute_virial) {
  return XpainnEvalFn::apply(pos, atomic_numbers, edge_index, ptr, cell, cell_offsets, prm, ip, fp, center_sorted, symmetric,
                             compute_forces, compute_virial);
}

// open-boundary neighbour list; one device-to-host read of the edge count, as the reference's nonzero()
std::tuple<Tensor, Tensor> radius_graph(const Tensor& pos_in, const Tensor& ptr, double cutoff) {
  need_hip(pos_in, "pos");
  const Tensor pos = pos_in.detach().contiguous();
  const Tensor ptr64 = ptr.to(at::kLong).contiguous();
  const int64_t N = pos.size(0), G = ptr64.numel() - 1;
  const int dt = dcode(pos);
  Tensor deg = i32(N, ptr64), rowptr = i32(N + 1, ptr64);
  void* st = cur_stream();
  XCALL(xeq_radius_graph_count(dt, pos.data_ptr(), (const int64_t*)ptr64.data_ptr(), G, N, cutoff, (int32_t*)deg.data_ptr(), st));
  const int64_t scan_bytes = xeq_exclusive_scan_i32_workspace(N);
  TORCH_CHECK(scan_bytes >= 0, "xeq::radius_graph: too many nodes");
  Tensor scan_work = at::empty({std::max<int64_t>(scan_bytes, 1)}, pos.options().dtype(at::kByte));
  XCALL(xeq_exclusive_scan_i32_ws((const int32_t*)deg.data_ptr(), N, (int32_t*)rowptr.data_ptr(), scan_work.data_ptr(), scan_bytes, st));
  const int64_t E = N > 0 ? (int64_t)rowptr[N].item<int32_t>() : 0;
  Tensor ei = at::empty({2, E}, ptr64.options());
  XCALL(xeq_radius_graph_fill(dt, pos.data_ptr(), (const int64_t*)ptr64.data_ptr(), G, N, cutoff, (const int32_t*)rowptr.data_ptr(), E,
                              (int64_t*)ei.data_ptr(), st));
  return {ei, rowptr.narrow(0, 0, N + 1)};
}

// periodic neighbour list of ONE system (data/radius_graph.py:195-275): positions are not wrapped, cell [3, 3], pbc [3].
// The twin of data.radius_graph.single_radius_graph of this package: the same host tables (xeq_pbc_image_counts /
// xeq_pbc_tables_host) and the same image-pruned search kernels, hence the same list bit for bit.  Two round trips, as there:
// the cell (9 numbers) comes to the host for the image counts, and the edge count sizes the output (the reference's nonzero()).
std::tuple<Tensor, Tensor, Tensor> radius_graph_pbc(const Tensor& pos_in, const Tensor& cell_in, const Tensor& pbc_in, double cutoff) {
  need_hip(pos_in, "pos");
  TORCH_CHECK(cell_in.dim() == 2 && cell_in.size(0) == 3 && cell_in.size(1) == 3, "xeq::radius_graph_pbc: cell must be [3, 3]");
  TORCH_CHECK(pbc_in.numel() == 3, "xeq::radius_graph_pbc: pbc must hold three flags");
  const Tensor pos = pos_in.detach().contiguous();
  const int dt = dcode(pos);
  const int64_t N = pos.size(0);
  const Tensor cell_h = cell_in.detach().to(pos.scalar_type()).to(at::kCPU).contiguous();   // round trip 1
  const Tensor pbc_h = pbc_in.detach().to(at::kCPU).to(at::kInt).reshape({3}).contiguous();
  const int32_t pbc[3] = {pbc_h.data_ptr<int32_t>()[0] != 0, pbc_h.data_ptr<int32_t>()[1] != 0, pbc_h.data_ptr<int32_t>()[2] != 0};
  int32_t reps[3];
  XCALL(xeq_pbc_image_counts(dt, cell_h.data_ptr(), 1, pbc, cutoff, reps));
  const int64_t nc = (int64_t)(2 * reps[0] + 1) * (2 * reps[1] + 1) * (2 * reps[2] + 1), n_tab = 6 * nc + 12;
  Tensor tab_h = at::empty({n_tab}, cell_h.options());
  XCALL(xeq_pbc_tables_host(dt, cell_h.data_ptr(), 1, reps, cutoff, tab_h.data_ptr(), n_tab));
  const Tensor tab = tab_h.to(pos.device());                                                 // one upload
  const Tensor grid = tab.narrow(0, 0, 3 * nc), offs = tab.narrow(0, 3 * nc, 3 * nc), recip = tab.narrow(0, 6 * nc, 9),
               thr = tab.narrow(0, 6 * nc + 9, 3);
  const Tensor ptr64 = at::tensor({(int64_t)0, N}, at::TensorOptions().dtype(at::kLong)).to(pos.device());
  const Tensor shift = at::zeros_like(pos);
  Tensor deg = i32(N, ptr64), rowptr = i32(N + 1, ptr64);
  void* st = cur_stream();
  XCALL(xeq_radius_graph_pbc_count_pruned(dt, pos.data_ptr(), (const int64_t*)ptr64.data_ptr(), 1, N, offs.data_ptr(), nc, cutoff,
                                          recip.data_ptr(), thr.data_ptr(), reps, (int32_t*)deg.data_ptr(), st));
  const int64_t scan_bytes = xeq_exclusive_scan_i32_workspace(N);
  TORCH_CHECK(scan_bytes >= 0, "xeq::radius_graph_pbc: too many nodes");
  Tensor scan_work = at::empty({std::max<int64_t>(scan_bytes, 1)}, pos.options().dtype(at::kByte));
  XCALL(xeq_exclusive_scan_i32_ws((const int32_t*)deg.data_ptr(), N, (int32_t*)rowptr.data_ptr(), scan_work.data_ptr(), scan_bytes, st));
  const int64_t E = N > 0 ? (int64_t)rowptr[N].item<int32_t>() : 0;                          // round trip 2
  Tensor ei = at::empty({2, E}, ptr64.options());
  Tensor cell_offsets = at::empty({E, 3}, pos.options());
  XCALL(xeq_radius_graph_pbc_fill_pruned(dt, pos.data_ptr(), (const int64_t*)ptr64.data_ptr(), 1, N, offs.data_ptr(), grid.data_ptr(),
                                         shift.data_ptr(), nc, cutoff, recip.data_ptr(), thr.data_ptr(), reps,
                                         (const int32_t*)rowptr.data_ptr(), E, (int64_t*)ei.data_ptr(), cell_offsets.data_ptr(), st));
  return {ei, cell_offsets, rowptr};
}

// ---------------------------------------------------------------------------------------------- linear layers of the training pass
// y = x W^T (+ b) as an autograd node whose reverse pass is written with itself and with XeqWGradFn (dL/dx = g W, dL/dW = g^T x), so
// every order of derivative stays on these two products, and every reduction over the N rows -- which the library's GEMMs run at a
// tenth of their rate for [576 x N] x [N x 128] and the like -- goes to xeq_wgrad (fp32).  Python twin: nn/training_ops.py LinearFn /
// WGradFn (same arithmetic; a C++ node costs the host a fifth of a Python one, which is what decides a host-launched training step).
struct XeqWGradFn;
Tensor xeq_wgrad_apply(const Tensor& a, const Tensor& b);

struct XeqLinearFn : public torch::autograd::Function<XeqLinearFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& W, const std::optional<Tensor>& b) {
    ctx->save_for_backward({x, W});
    const bool has_bias = b.has_value() && b->defined();
    ctx->saved_data["has_bias"] = has_bias;
    return has_bias ? at::addmm(*b, x, W.t()) : at::mm(x, W.t());
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &W = saved[1], &go = g[0];
    Tensor dx, dW, db;
    if (ctx->needs_input_grad(0)) dx = XeqLinearFn::apply(go, W.t(), std::optional<Tensor>());
    if (ctx->needs_input_grad(1)) dW = xeq_wgrad_apply(go, x);
    if (ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2)) db = go.sum(0);
    return {dx, dW, db};
  }
};

struct XeqWGradFn : public torch::autograd::Function<XeqWGradFn> {   // a^T b over the rows: a [N, M], b [N, K] -> [M, K]
  static Tensor forward(AutogradContext* ctx, const Tensor& a, const Tensor& b) {
    ctx->save_for_backward({a, b});
    if (!(a.is_cuda() && a.scalar_type() == at::kFloat && b.scalar_type() == at::kFloat && a.dim() == 2 && b.dim() == 2 && a.size(0) == b.size(0)))
      return at::mm(a.t(), b);
    const Tensor ac = a.stride(1) == 1 ? a : a.contiguous(), bc = b.stride(1) == 1 ? b : b.contiguous();
    const int64_t n = ac.size(0);
    const int M = (int)ac.size(1), K = (int)bc.size(1);
    const int chunks = xeq_wgrad_chunks(n, M, K);
    Tensor parts = at::empty({chunks, (int64_t)M * K}, ac.options());
    XCALL(xeq_wgrad(ac.data_ptr(), ac.stride(0), bc.data_ptr(), bc.stride(0), n, M, K, 0, chunks, parts.data_ptr(), cur_stream()));
    return (chunks > 1 ? parts.sum(0) : parts[0]).view({M, K});
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &a = saved[0], &b = saved[1], &U = g[0];
    Tensor da, db;
    if (ctx->needs_input_grad(0)) da = XeqLinearFn::apply(b, U, std::optional<Tensor>());       // b U^T
    if (ctx->needs_input_grad(1)) db = XeqLinearFn::apply(a, U.t(), std::optional<Tensor>());   // a U
    return {da, db};
  }
};
Tensor xeq_wgrad_apply(const Tensor& a, const Tensor& b) { return XeqWGradFn::apply(a, b); }

Tensor linear_op(const Tensor& x, const Tensor& W, const std::optional<Tensor>& b) {
  TORCH_CHECK(x.dim() == 2 && W.dim() == 2 && x.size(1) == W.size(1), "xeq::linear: x [N, K], W [M, K]");
  return XeqLinearFn::apply(x, W, b);
}

}  // namespace

TORCH_LIBRARY(xeq, m) {
  m.def(
      "xpainn_eval(Tensor pos, Tensor atomic_numbers, Tensor edge_index, Tensor ptr, Tensor? cell, Tensor? cell_offsets, "
      "Tensor[] params, int[] iparams, float[] fparams, bool center_sorted, bool symmetric, bool compute_forces, "
      "bool compute_virial) -> Tensor[]");
  m.def("radius_graph(Tensor pos, Tensor ptr, float cutoff) -> (Tensor, Tensor)");
  m.def("radius_graph_pbc(Tensor pos, Tensor cell, Tensor pbc, float cutoff) -> (Tensor, Tensor, Tensor)");
  m.def("linear(Tensor x, Tensor W, Tensor? b) -> Tensor", linear_op);   // differentiable to every order (XeqLinearFn / XeqWGradFn)
}

TORCH_LIBRARY_IMPL(xeq, Autograd, m) { m.impl("xpainn_eval", xpainn_eval); }
TORCH_LIBRARY_IMPL(xeq, CompositeExplicitAutograd, m) {
  m.impl("radius_graph", radius_graph);
  m.impl("radius_graph_pbc", radius_graph_pbc);
}
