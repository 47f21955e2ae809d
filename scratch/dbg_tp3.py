import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd import tp
torch.manual_seed(0)
cases = {"p0": [(0, 0, 0, "uvv", True, 1.0)], "p1": [(0, 1, 1, "uvv", True, 1.0)], "both": [(0, 0, 0, "uvv", True, 1.0), (0, 1, 1, "uvv", True, 1.0)]}
for name, ins in cases.items():
    mod = tp.TensorProduct("6x0e", "3x0e+5x1o", "3x0e+5x1o", ins, internal_weights=True, shared_weights=True).double().cuda()
    x = torch.randn(4, 6, dtype=torch.float64, device="cuda", requires_grad=True); y = torch.randn(4, 18, dtype=torch.float64, device="cuda", requires_grad=True)
    r = {}
    for fused in (False, True):
        mod.fused = fused
        o = mod(x, y)
        g = torch.cos(torch.arange(o.numel(), device="cuda", dtype=torch.float64)).reshape(o.shape)
        r[fused] = [o.detach()] + list(torch.autograd.grad(o, [x, y, mod.weight], g))
    print(name, [f"{float((a - b).abs().max()):.2e}" for a, b in zip(r[True], r[False])], "gx per-path", r[False][1][0].cpu().numpy().round(3), "fused", r[True][1][0].cpu().numpy().round(3))
ins = cases["both"]
mod = tp.TensorProduct("6x0e", "3x0e+5x1o", "3x0e+5x1o", ins, internal_weights=True, shared_weights=True).double().cuda()
x = torch.randn(4, 6, dtype=torch.float64, device="cuda", requires_grad=True); y = torch.randn(4, 18, dtype=torch.float64, device="cuda", requires_grad=True)
W = mod.weight.detach().clone()
for label, sel in (("only path 0 weights", slice(18, 48)), ("only path 1 weights", slice(0, 18))):
    with torch.no_grad():
        mod.weight.copy_(W); mod.weight[sel] = 0
    r = {}
    for fused in (False, True):
        mod.fused = fused
        o = mod(x, y)
        g = torch.cos(torch.arange(o.numel(), device="cuda", dtype=torch.float64)).reshape(o.shape)
        r[fused] = torch.autograd.grad(o, x, g)[0]
    print(label, r[False][0].cpu().numpy().round(3), r[True][0].cpu().numpy().round(3))
import math
with torch.no_grad():
    mod.weight.copy_(W); mod.weight[0:18] = 0
W1 = mod.weight.detach()[18:48].reshape(6, 5)
o = mod(x, y)
g = torch.cos(torch.arange(o.numel(), device="cuda", dtype=torch.float64)).reshape(o.shape)
g1 = g[:, 3:18].reshape(4, 5, 3); y1 = y.detach()[:, 3:18].reshape(4, 5, 3)
c1 = mod.coeffs[1]
ref = c1 / math.sqrt(3) * torch.einsum("uv,nvk,nvk->nu", W1, g1, y1)
print("closed form", ref[0].cpu().numpy().round(3), "coeffs", mod.coeffs)
t = mod._fused_table("dx1", x)
wt, stride = mod._pass_weights(t, mod.weight.detach())
print("paths", [tuple(t["paths"][10 * i: 10 * i + 10]) for i in range(t["n_paths"])], "cg_off", list(t["cg_off"]), "w_off", list(t["w_off"]), "coeff", list(t["coeff"]), "stride", stride)
print("cg", t["cg"].cpu().numpy().round(3), "wt[18:24]", wt[18:24].cpu().numpy().round(3), "W1.t()[0]", W1.t()[0].cpu().numpy().round(3))
# emulate
G = g.cpu().numpy(); Y = y.detach().cpu().numpy(); cgn = t["cg"].cpu().numpy(); wn = wt.cpu().numpy()
import numpy as np
out = np.zeros(6)
for i in range(t["n_paths"]):
    o1_, o2_, oo_, m1, m2, mo, l1, l2, l3, md = t["paths"][10 * i: 10 * i + 10]
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    C = cgn[t["cg_off"][i]: t["cg_off"][i] + d1 * d2 * d3].reshape(d1, d2, d3)
    Wp = wn[t["w_off"][i]: t["w_off"][i] + m1 * mo].reshape(m1, mo)
    a = G[0, o1_: o1_ + m1 * d1].reshape(m1, d1); b = Y[0, o2_: o2_ + m2 * d2].reshape(m2, d2)
    z = np.einsum("ijk,ui,uj->uk", C, a, b)[:, 0]
    out += t["coeff"][i] * (Wp * z[:, None]).sum(0)
print("emulated fused", out.round(3))
def emu(cg_shift=None, w_shift=None, coeff_idx=None, mul_from=None):
    out = np.zeros(6)
    for i in range(t["n_paths"]):
        o1_, o2_, oo_, m1, m2, mo, l1, l2, l3, md = t["paths"][10 * i: 10 * i + 10]
        d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
        co = t["cg_off"][i] if cg_shift is None else cg_shift
        C = np.resize(cgn[co:], d1 * d2 * d3).reshape(d1, d2, d3) if co + d1 * d2 * d3 > len(cgn) else cgn[co: co + d1 * d2 * d3].reshape(d1, d2, d3)
        wo = t["w_off"][i] if w_shift is None else w_shift
        Wp = wn[wo: wo + m1 * mo].reshape(m1, mo)
        a = G[0, o1_: o1_ + m1 * d1].reshape(m1, d1); b = Y[0, o2_: o2_ + m2 * d2].reshape(m2, d2)
        z = np.einsum("ijk,ui,uj->uk", C, a, b)[:, 0]
        cf = t["coeff"][i] if coeff_idx is None else t["coeff"][coeff_idx]
        out += cf * (Wp * z[:, None]).sum(0)
    return out.round(3)
print("gpu fused (wrong)", r[True][0].cpu().numpy().round(3))
print("cg_off ignored", emu(cg_shift=0)); print("coeff of path 0", emu(coeff_idx=0))
# fp32 coefficient / struct misread hypotheses
out = np.zeros(6)
for i in range(t["n_paths"]):
    o1_, o2_, oo_, m1, m2, mo, l1, l2, l3, md = t["paths"][10 * i: 10 * i + 10]
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    C = cgn[t["cg_off"][i]: t["cg_off"][i] + d1 * d2 * d3].reshape(d1, d2, d3)
    Wp = wn[t["w_off"][i]: t["w_off"][i] + m1 * mo].reshape(m1, mo)
    a = G[0, o1_: o1_ + m1 * d1].reshape(m1, d1); b = Y[0, o2_: o2_ + m2 * d2].reshape(m2, d2)
    z = np.einsum("ijk,ui,uj->uk", C, a, b)[:, 0]
    m_eff = 3   # multiplicity of path 0 used for every path
    out += t["coeff"][i] * (Wp[:m_eff] * z[:m_eff, None]).sum(0)
print("mul1 of path 0 for all", out.round(3))
buf = torch.zeros(4, 6, dtype=torch.float64, device="cuda")
mod._run_fused(t, g.contiguous(), y.detach().contiguous(), mod.weight.detach(), buf)
torch.cuda.synchronize()
print("direct _run_fused", buf[0].cpu().numpy().round(3))
os.environ["XEQ_TP_GENERIC"] = "1"
buf.zero_(); mod._run_fused(t, g.contiguous(), y.detach().contiguous(), mod.weight.detach(), buf); torch.cuda.synchronize()
print("direct generic   ", buf[0].cpu().numpy().round(3))
