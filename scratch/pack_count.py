"""Which path re-packs MLP weights: run under rocprofv3 --kernel-trace --stats with MODE=native|eager|replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd import runtime
dev = "cuda"
mode = os.environ.get("MODE", "native")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
pos, z, ptr, _ = syn.make_workload("qm9_1024", 1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
transform = NeighborTransform(model.cutoff_radius)
if mode == "native":
    from xequinet_amd.interface.scripted import XPaiNNNative
    native = XPaiNNNative(model)
    def step():
        b = transform(XequiBatch(pos_d.detach(), z_d, ptr_d))
        return native(b.pos, b.atomic_numbers, b.edge_index, b.ptr, None, None, True, True, True, False)
elif mode == "eager":
    def step():
        b = transform(XequiBatch(pos_d.detach(), z_d, ptr_d))
        with torch.enable_grad():
            return model(b.to_dict(), compute_forces=True, compute_virial=False)
else:
    g = runtime.GraphedModel(model, compute_forces=True, compute_virial=False, tune_gemms=False)
    def step():
        b = transform(XequiBatch(pos_d.detach(), z_d, ptr_d))
        return g(b.to_dict())
for _ in range(10):
    step()
torch.cuda.synchronize()
print("done", mode)
