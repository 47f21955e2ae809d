import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_tp import _case
mod, in1, in2, out, ins, x, y, w, _ = _case("uvv", True, True, n=37)
mod = mod.to("cuda")
xs = [t.cuda().requires_grad_() for t in (x, y)]
o = mod(*xs)
g = torch.ones_like(o) * 0.37 + torch.arange(o.shape[1], device="cuda") * 0.01
gx = torch.autograd.grad(o, xs[0], g, retain_graph=True)[0]
print("tied   ", gx[0, :6].cpu().numpy().round(4))
os.environ["XEQ_TP_GENERIC"] = "1"
gx2 = torch.autograd.grad(o, xs[0], g)[0]
print("generic", gx2[0, :6].cpu().numpy().round(4))
mod.fused = False
o3 = mod(*xs)
gx3 = torch.autograd.grad(o3, xs[0], g)[0]
print("per-path", gx3[0, :6].cpu().numpy().round(4), "out diff", float((o3 - o).abs().max()))
d = (gx - gx3).abs()
print("max diff", float(d.max()), "nodes", torch.nonzero(d.amax(1) > 1e-9).flatten().tolist()[:10], "cols", torch.nonzero(d.amax(0) > 1e-9).flatten().tolist()[:40])
t = mod._fused_table("dx1", xs[0])
print([tuple(t["paths"][10 * i: 10 * i + 10]) for i in range(t["n_paths"])][:8], list(t["w_off"])[:8])
import numpy as np
G = g.double().cpu().numpy(); Y = y.double().numpy()
wt, _ = mod._pass_weights(t, mod.weight.detach())
wt = wt.cpu().double().numpy(); cgs = t["cg"].cpu().double().numpy()
acc = np.zeros(6)
for i in range(t["n_paths"]):
    o1_, o2_, oo_, m1, m2, mo, l1, l2, l3, md = t["paths"][10 * i: 10 * i + 10]
    if oo_ != 0: continue
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    C = cgs[t["cg_off"][i]: t["cg_off"][i] + d1 * d2 * d3].reshape(d1, d2, d3)
    W = wt[t["w_off"][i]: t["w_off"][i] + m1 * mo].reshape(m1, mo)
    a = G[0, o1_: o1_ + m1 * d1].reshape(m1, d1); b = Y[0, o2_: o2_ + m2 * d2].reshape(m2, d2)
    z = np.einsum("ijk,ui,uj->uk", C, a, b)
    contrib = t["coeff"][i] * np.einsum("uw,uk->wk", W, z)[:, 0]
    print("  path", i, "mul1", m1, contrib.round(4))
    acc += contrib
print("numpy slot 0:", acc.round(4))
