"""Host / device time of the per-step plumbing in front of the model: XequiBatch construction and NeighborTransform."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
dev = "cuda"
pos, z, ptr, _ = syn.make_workload("qm9_1024", 1234)
pos_d, z_d, ptr_d = torch.tensor(pos, dtype=torch.float32, device=dev), torch.tensor(z, device=dev), torch.tensor(ptr, device=dev)
z32 = z_d.to(torch.int32)
tr = NeighborTransform(5.0)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("XequiBatch(z int64): %.1f us" % t(lambda: XequiBatch(pos_d, z_d, ptr_d)))
print("XequiBatch(z int32): %.1f us" % t(lambda: XequiBatch(pos_d, z32, ptr_d)))
print("XequiBatch + transform: %.1f us" % t(lambda: tr(XequiBatch(pos_d, z32, ptr_d))))
b = XequiBatch(pos_d, z32, ptr_d)
def only_tr():
    b.edge_index = None
    return tr(b)
print("transform alone: %.1f us" % t(only_tr))
