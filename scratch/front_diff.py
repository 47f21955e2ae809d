import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import resolve_model
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.interface.scripted import XPaiNNNative
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(torch.float32).to(dev)
native = XPaiNNNative(model)
t = lambda a, dt=None: torch.as_tensor(a, device=dev).to(dt) if dt is not None else torch.as_tensor(a, device=dev)
for n_mol in [int(a) for a in sys.argv[1:]] or [24, 400, 1024]:
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=13)
    b = NeighborTransform(5.0)(XequiBatch(t(pos, torch.float32), t(z), t(ptr)))
    data = b.to_dict()
    with torch.enable_grad():
        want = model(dict(data), compute_forces=True, compute_virial=False)
    got = native(data["pos"].detach(), data["atomic_numbers"], data["edge_index"], data["ptr"], None, None, True, True, True, False)
    with torch.enable_grad():
        want2 = model(dict(data), compute_forces=True, compute_virial=False)
    got2 = native(data["pos"].detach(), data["atomic_numbers"], data["edge_index"], data["ptr"], None, None, True, True, True, False)
    d = (got[2] - want["forces"].detach()).abs()
    print(n_mol, "atoms", len(pos), "forces equal", torch.equal(got[2], want["forces"].detach()), "max diff", float(d.max()), "rows differing", int((d.amax(1) > 0).sum()),
          "| python repeats", torch.equal(want["forces"], want2["forces"]), "native repeats", torch.equal(got[2], got2[2]), "| energy equal", torch.equal(got[0], want["energy"].detach()),
          "atomic equal", torch.equal(got[1], want["atomic_energies"].detach()), flush=True)
