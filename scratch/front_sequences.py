"""The launch sequences of the two fronts (Python modules / xeq::xpainn_eval) on the same batch, side by side.
python scratch/front_sequences.py [n_molecules]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import lib, keys, ops
from xequinet_amd.nn import resolve_model
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.interface.scripted import XPaiNNNative

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(torch.float32).to(dev)
native = XPaiNNNative(model)
n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=13)
t = lambda a, dt=None: torch.as_tensor(a, device=dev).to(dt) if dt is not None else torch.as_tensor(a, device=dev)
b = NeighborTransform(5.0)(XequiBatch(t(pos, torch.float32), t(z), t(ptr)))
data = b.to_dict()

def py():
    d = dict(data)      # a fresh EdgeGraph with the builder's promises (center-sorted, symmetric): this front rebuilds the views and plans, as the operator does
    d[keys.EDGE_GRAPH] = ops.EdgeGraph(data["edge_index"], data["pos"].shape[0], center_sorted=True, ptr=data["ptr"], symmetric=True)
    with torch.enable_grad():
        return model(d, compute_forces=True, compute_virial=False)
def cc():
    return native(data["pos"].detach(), data["atomic_numbers"], data["edge_index"], data["ptr"], None, None, True, True, True, False)
for _ in range(2): py(); cc()
torch.cuda.synchronize()
c0 = lib.launch_count(); want = py(); a = lib.launch_names(c0)
c0 = lib.launch_count(); got = cc(); bb = lib.launch_names(c0)
print("bitwise equal forces:", torch.equal(got[2], want["forces"].detach()), " energy:", torch.equal(got[0], want["energy"].detach()))
print(f"python modules: {len(a)} launches; registered operator: {len(bb)} launches; same sequence: {a == bb}")
for i in range(max(len(a), len(bb))):
    x = a[i] if i < len(a) else "-"; y = bb[i] if i < len(bb) else "-"
    print(f"{i:3d}  {x:38s} {y:38s} {'' if x == y else '<<<'}")
