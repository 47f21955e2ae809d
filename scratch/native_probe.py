"""Where the registered operator's wall time goes: host enqueue time vs GPU time, against the Python modules."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.interface.scripted import XPaiNNNative
dev = "cuda"
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
pos, z, ptr, _ = syn.make_workload("qm9_1024", 1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
transform = NeighborTransform(model.cutoff_radius)
native = XPaiNNNative(model)
def nl():
    return transform(XequiBatch(pos_d.detach(), z_d, ptr_d))
def step_native(batch=None):
    b = batch or nl()
    return native(b.pos, b.atomic_numbers, b.edge_index, b.ptr, None, None, True, True, True, False)
def step_eager(batch=None):
    b = batch or nl()
    with torch.enable_grad():
        return model(b.to_dict(), compute_forces=True, compute_virial=False)
def measure(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h0 = time.perf_counter(); fn(); host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3
for name, fn in (("neighbour list only", nl), ("native op incl. list", step_native), ("python modules incl. list", step_eager)):
    w, h = measure(fn)
    print(f"{name}: wall {w:.3f} ms, host time inside the call {h:.3f} ms")
b = nl(); torch.cuda.synchronize()
for name, fn in (("native op, list given", lambda: step_native(b)), ("python modules, list given", lambda: step_eager(b))):
    w, h = measure(fn)
    print(f"{name}: wall {w:.3f} ms, host time inside the call {h:.3f} ms")
