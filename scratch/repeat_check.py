"""Are whole evaluations bitwise repeatable at sizes that put two or more waves of every kernel on a SIMD?  (a sporadic hazard shows as
run-to-run differences)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import resolve_model
from xequinet_amd.data import synthetic as syn, NeighborTransform, XequiBatch
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(torch.float32).to(dev)
for wl in sys.argv[1:] or ["qm9_1024", "md17_4096"]:
    pos, z, ptr, _ = syn.make_workload(wl, seed=1234)
    t = lambda a, dt=None: torch.as_tensor(a, device=dev) if dt is None else torch.as_tensor(a, device=dev).to(dt)
    b = NeighborTransform(5.0)(XequiBatch(t(pos, torch.float32), t(z), t(ptr)))
    ref = None; bad = 0
    for rep in range(12):
        with torch.enable_grad():
            out = model(b.to_dict(), compute_forces=True)
        E, F = out["energy"].detach().clone(), out["forces"].detach().clone()
        if ref is None: ref = (E, F)
        elif not (torch.equal(E, ref[0]) and torch.equal(F, ref[1])):
            bad += 1
            d = (F - ref[1]).abs()
            print(f"  {wl} run {rep}: {int((d.amax(1) > 0).sum())} atoms differ, max {float(d.max()):.3e}")
    print(f"{wl}: {bad} of 11 repeats differ from the first run")
