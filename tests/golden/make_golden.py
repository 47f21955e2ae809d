"""Generate golden fixtures by RUNNING THE REFERENCE in the build container.

Run once, here (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

Only the e3nn-free reference modules can be imported (e3nn / torch_scatter /
torch_cluster / torch_geometric are absent and pip is offline), through the
container-only shims of SURVEY.md Appendix C:

* fake ``torch_scatter`` (``scatter``/``scatter_sum`` via ``zeros.index_add``),
* fake ``xequinet.utils`` exposing ``get_embedding_tensor`` (mirrors utils/qc.py:232-237),
* an empty ``xequinet.nn`` namespace package so nn/__init__.py (which pulls e3nn) is bypassed.

The outputs are *data* (inputs + expected outputs, .npz); no reference source is
copied.  Fixtures keep every pair >= 1e-4 A away from the cutoff.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import xpainn_oracle as orc  # noqa: E402  (synthetic input generators only)
from xequinet_amd.data import synthetic as syn


def _install_shims():
    sys.path.insert(0, REF)
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, reduce="sum", **kw):
        assert reduce in ("sum", "add")
        shape = list(src.shape)
        shape[dim] = int(index.max()) + 1
        return torch.zeros(shape, dtype=src.dtype).index_add(dim, index, src)

    ts.scatter = scatter
    ts.scatter_sum = lambda src, index, dim=0, **kw: scatter(src, index, dim)
    sys.modules["torch_scatter"] = ts
    import xequinet  # noqa: F401  (real; only reads LOCAL_RANK)

    utils = types.ModuleType("xequinet.utils")

    def get_embedding_tensor(embed_basis="gfn2-xtb", aux_basis="aux28"):
        d = torch.load(f"{REF}/xequinet/utils/pre_computed/{embed_basis}_{aux_basis}.pt")
        t = torch.stack(list(d.values()))
        t = torch.cat([torch.zeros(1, t.shape[-1]), t])
        return t.to(torch.get_default_dtype())

    utils.get_embedding_tensor = get_embedding_tensor
    sys.modules["xequinet.utils"] = utils
    nn_ns = types.ModuleType("xequinet.nn")
    nn_ns.__path__ = [f"{REF}/xequinet/nn"]
    sys.modules["xequinet.nn"] = nn_ns
    return utils


def _by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _away_from_cutoff(pos, cutoff, ptr=None, tol=1e-4):
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    return np.all(np.abs(d - cutoff) > tol)


def main():
    torch.manual_seed(0)
    utils = _install_shims()
    basic = importlib.import_module("xequinet.nn.basic")
    rbf = importlib.import_module("xequinet.nn.rbf")
    painn = importlib.import_module("xequinet.nn.painn")
    rg = _by_path("ref_radius_graph", f"{REF}/xequinet/data/radius_graph.py")
    keys = importlib.import_module("xequinet.keys")

    # ---- 1. radial bases / envelopes on a fixed grid (nn/rbf.py) ------------------
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dt)
        d = torch.cat([torch.linspace(0.3, 6.0, 58), torch.tensor([0.5, 2.0, 4.999, 5.0, 5.0001])]).to(dt).view(-1, 1)
        out = {
            "dist": d.numpy(),
            "bessel20_rc5": rbf.SphericalBesselj0(20, 5.0)(d).detach().numpy(),
            "bessel8_rc4": rbf.SphericalBesselj0(8, 4.0)(d).detach().numpy(),
            "cosine_rc5": rbf.CosineCutoff(5.0)(d).numpy(),
            "poly3_rc5": rbf.PolynomialCutoff(5.0)(d).numpy(),
            "gauss20_rc5": rbf.GaussianSmearing(20, 5.0)(d).detach().numpy(),
        }
        np.savez_compressed(os.path.join(HERE, f"rbf_{tag}.npz"), **out)
    torch.set_default_dtype(torch.float32)

    # ---- 2. embedding table rows (utils/qc.py:222-237) ---------------------------
    zs = np.array([0, 1, 6, 7, 8, 9, 17, 35, 86])
    np.savez_compressed(
        os.path.join(HERE, "embed_rows.npz"),
        z=zs,
        aux56=utils.get_embedding_tensor("gfn2-xtb", "aux56")[zs].double().numpy(),
        aux28=utils.get_embedding_tensor("gfn2-xtb", "aux28")[zs].double().numpy(),
    )

    # ---- 3. radius_graph_pbc (data/radius_graph.py:35-192) -----------------------
    cases = {}
    # (a) 192-atom water-density cubic box
    pos, z, ptr, cell = syn.synth_water_box(4, seed=5)
    cases["water192"] = (pos, np.array([192]), cell, [True, True, True], 5.0)
    # (b) two graphs, different cubic cells, un-wrapped positions (exercises :186-190)
    p1, _, _, c1 = syn.synth_water_box(3, seed=6)
    p2, _, _, c2 = syn.synth_water_box(2, seed=7)
    p1 = p1 + np.array([13.0, -7.5, 0.3])  # push outside the cell
    cases["two_graphs_unwrapped"] = (np.concatenate([p1, p2]), np.array([len(p1), len(p2)]),
                                     np.concatenate([c1, c2]), [True, True, True], 4.0)
    # (c) triclinic cell
    rng = np.random.default_rng(3)
    tri = np.array([[9.0, 0.0, 0.0], [2.5, 8.0, 0.0], [1.0, 2.0, 7.5]])
    p3 = rng.uniform(0, 1, size=(40, 3)) @ tri
    cases["triclinic40"] = (p3, np.array([40]), tri[None], [True, True, True], 4.5)
    # (d) slab: periodic in x,y only
    p4 = rng.uniform(0, 1, size=(30, 3)) @ (np.eye(3) * np.array([6.0, 7.0, 20.0]))
    cases["slab30"] = (p4, np.array([30]), (np.eye(3) * np.array([6.0, 7.0, 20.0]))[None], [True, True, False], 5.0)
    for name, (pos, npg, cell, pbc, rc) in cases.items():
        pos32 = torch.tensor(pos, dtype=torch.float32)
        cell32 = torch.tensor(cell, dtype=torch.float32)
        pbc_t = torch.tensor([pbc] * len(npg))
        ei, co = rg.radius_graph_pbc(pos=pos32, n_nodes_per_graph=torch.tensor(npg), pbc=pbc_t, cell=cell32, cutoff=rc)
        # keep the fixture only if no pair sits within 1e-4 of the cutoff / 0.01 thresholds
        np.savez_compressed(os.path.join(HERE, f"radius_graph_pbc_{name}.npz"), pos=pos32.numpy(), n_per_graph=npg,
                 cell=cell32.numpy(), pbc=np.array(pbc), cutoff=rc, edge_index=ei.numpy(), cell_offsets=co.numpy())
        print(name, "edges", ei.shape[1], "per atom %.2f" % (ei.shape[1] / len(pos)))
    # single_radius_graph (TorchScript variant, :195-275) on the cubic box
    pos, z, ptr, cell = syn.synth_water_box(4, seed=5)
    ei, co = rg.single_radius_graph(torch.tensor(pos, dtype=torch.float32), torch.tensor([True, True, True]),
                                    torch.tensor(cell[0], dtype=torch.float32), 5.0)
    np.savez_compressed(os.path.join(HERE, "single_radius_graph_water192.npz"), edge_index=ei.numpy(), cell_offsets=co.numpy())

    # ---- 4. compute_edge_data (nn/basic.py:60-140) --------------------------------
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dt)
        # (a) aspirin-like, non-PBC, canonical all-pairs-within-5A graph
        pos, z, ptr = syn.synth_aspirin()
        ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
        data = {keys.POSITIONS: torch.tensor(pos, dtype=dt), keys.EDGE_INDEX: torch.tensor(ei)}
        out = basic.compute_edge_data(data, compute_forces=False)
        np.savez_compressed(os.path.join(HERE, f"edge_data_aspirin_{tag}.npz"), pos=pos, edge_index=ei,
                 edge_vector=out[keys.EDGE_VECTOR].numpy(), edge_length=out[keys.EDGE_LENGTH].numpy())
        # (b) two-graph PBC batch using the reference's own graph
        f = np.load(os.path.join(HERE, "radius_graph_pbc_two_graphs_unwrapped.npz"))
        npg = f["n_per_graph"]
        batch = np.repeat(np.arange(len(npg)), npg)
        data = {
            keys.POSITIONS: torch.tensor(f["pos"], dtype=dt),
            keys.EDGE_INDEX: torch.tensor(f["edge_index"]),
            keys.CELL: torch.tensor(f["cell"], dtype=dt),
            keys.CELL_OFFSETS: torch.tensor(f["cell_offsets"], dtype=dt),
            keys.BATCH: torch.tensor(batch),
            keys.BATCH_PTR: torch.tensor(np.concatenate([[0], np.cumsum(npg)])),
        }
        out = basic.compute_edge_data(data, compute_forces=False)
        np.savez_compressed(os.path.join(HERE, f"edge_data_pbc2_{tag}.npz"),
                 edge_vector=out[keys.EDGE_VECTOR].numpy(), edge_length=out[keys.EDGE_LENGTH].numpy())
    torch.set_default_dtype(torch.float32)

    # ---- 5. PaiNN twin (nn/painn.py, e3nn-free; same dataflow as XPaiNN at l<=1) ----
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(123)
    F_ = 16
    emb = painn.Embedding(node_dim=F_, num_basis=8, embed_basis="gfn2-xtb", aux_basis="aux56", cutoff=4.0)
    blocks = [(painn.PainnMessage(F_, 8), painn.PainnUpdate(F_)) for _ in range(2)]
    w_out = torch.randn(F_)
    pos, z, ptr = syn.synth_qm9_batch(3, seed=99)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 4.0)
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    data = {
        keys.POSITIONS: torch.tensor(pos),
        keys.ATOMIC_NUMBERS: torch.tensor(z.astype(np.int64)),
        keys.EDGE_INDEX: torch.tensor(ei),
        keys.BATCH: torch.tensor(batch),
        keys.BATCH_PTR: torch.tensor(ptr),
    }
    data = basic.compute_edge_data(data, compute_forces=True)
    data = emb(data)
    for m, u in blocks:
        data = u(m(data))
    e_atom = data[keys.NODE_INVARIANT] @ w_out
    energy = torch.zeros(len(ptr) - 1).index_add(0, data[keys.BATCH], e_atom)
    data[keys.TOTAL_ENERGY] = energy
    res = basic.compute_properties(data, compute_forces=True, training=False)
    sd = {"emb." + k: v.detach().numpy() for k, v in emb.state_dict().items()}
    for i, (m, u) in enumerate(blocks):
        sd.update({f"message_{i}." + k: v.detach().numpy() for k, v in m.state_dict().items()})
        sd.update({f"update_{i}." + k: v.detach().numpy() for k, v in u.state_dict().items()})
    np.savez_compressed(os.path.join(HERE, "painn_twin_f64.npz"), pos=pos, z=z, ptr=ptr, edge_index=ei, w_out=w_out.numpy(),
             node_dim=F_, num_basis=8, cutoff=4.0, energy=energy.detach().numpy(),
             forces=res[keys.FORCES].detach().numpy(),
             node_invariant=data[keys.NODE_INVARIANT].detach().numpy(),
             node_equivariant=data[keys.NODE_EQUIVARIANT].detach().numpy(), **sd)
    print("painn twin energy", energy.detach().numpy(), "sum F", res[keys.FORCES].sum(0).detach().numpy())
    torch.set_default_dtype(torch.float32)


if __name__ == "__main__":
    main()
