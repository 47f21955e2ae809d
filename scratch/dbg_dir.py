import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_training import _model, _batch, _targets, SMALL, _Frozen, DEV
from xequinet_amd import keys, train
host, dev = _batch(6, 5, torch.float64)
tgt = {k: v.to(DEV) for k, v in _targets(host, 7, False).items()}
for w in ({keys.FORCES: 10.0}, {keys.TOTAL_ENERGY: 1.0}, {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 10.0}):
    a = _model(torch.float64, **SMALL).train(); b = _model(torch.float64, **SMALL).train()
    la, _ = train.train_step_directional(a, dict(dev), tgt, _Frozen(a.parameters()), w, order=4)
    lb, _ = train.train_step(b, dict(dev), tgt, _Frozen(b.parameters()), w)
    print("weights", w, "loss", la.item(), lb.item())
    for (n, p), (_, q) in list(zip(a.named_parameters(), b.named_parameters()))[:6]:
        if q.grad is None: print(n, "no ref grad", None if p.grad is None else p.grad.abs().max().item()); continue
        pg = p.grad if p.grad is not None else torch.zeros_like(q.grad)
        cos = (pg * q.grad).sum() / (pg.norm() * q.grad.norm() + 1e-300)
        print(f"  {n:40s} |dir| {pg.norm():.3e} |exact| {q.grad.norm():.3e} cos {cos:.6f}")
print("---- pieces")
m = _model(torch.float64, **SMALL).train()
pos = dev["pos"].detach()
for p in m.parameters(): p.requires_grad_(False)
with torch.enable_grad():
    out = m(dict(dev), True, False)
F = out[keys.FORCES].detach(); 
for p in m.parameters(): p.requires_grad_(True)
v = torch.randn_like(F) * 0.01
h = 1e-3
def E_at(pp_):
    d = dict(dev); d[keys.POSITIONS] = pp_
    return m(d, False, False)[keys.TOTAL_ENERGY].sum()
ep, em = E_at(pos + h * v), E_at(pos - h * v)
print("D_v E numeric", ((ep - em) / (2 * h)).item(), " analytic -(v.F)", -(v * F).sum().item())
# exact: gradient of -(v . F) w.r.t. a parameter through the differentiable form
m2 = _model(torch.float64, **SMALL).train()
o2 = m2(dict(dev), True, False)
gex = torch.autograd.grad(-(v * o2[keys.FORCES]).sum() * -1.0, [m2.mods["embedding"].embedding[1].weight], allow_unused=True)[0]   # d(v.F)/dtheta
m.zero_grad()
(-(ep - em) / (2 * h)).backward()
gd = m.mods["embedding"].embedding[1].weight.grad
print("d(v.F)/dtheta: directional norm", gd.norm().item(), "exact norm", gex.norm().item(), "cos", ((gd * gex).sum() / (gd.norm() * gex.norm())).item())
print("keys", list(dev.keys()))
m.eval()
with torch.enable_grad():
    d = dict(dev); d[keys.POSITIONS] = (pos + h * v)
    e_inf = m(d, True, False)[keys.TOTAL_ENERGY].sum().item()
m.train()
print("E(pos+hv): native training pass", ep.item(), " inference path", e_inf, " E(pos)", E_at(pos).item())
from oracle import xpainn_oracle as orc
sd = {k: v_.detach().cpu().double() for k, v_ in m.state_dict().items()}
hh = dict(host); hh["pos"] = (pos + h * v).cpu()
print("oracle at pos+hv (same list)", orc.XPaiNNOracle(sd, **SMALL)(hh, False, False)["energy"].sum().item())
W = m.mods["embedding"].embedding[1].weight
def vF():
    for p in m.parameters(): p.requires_grad_(False)
    with torch.enable_grad():
        f = m(dict(dev), True, False)[keys.FORCES].detach()
    for p in m.parameters(): p.requires_grad_(True)
    return (v * f).sum().item()
for (i, j) in ((0, 0), (3, 5), (10, 20)):
    d_ = 1e-5
    with torch.no_grad(): W[i, j] += d_
    a_ = vF()
    with torch.no_grad(): W[i, j] -= 2 * d_
    b_ = vF()
    with torch.no_grad(): W[i, j] += d_
    print(f"W[{i},{j}]: finite difference in theta {(a_ - b_) / (2 * d_):+.6e}   directional {gd[i, j].item():+.6e}   double backward {gex[i, j].item():+.6e}")
def gradE(model_, pp_, native):
    model_.native_training = native
    model_.zero_grad()
    d = dict(dev); d[keys.POSITIONS] = pp_
    model_(d, False, False)[keys.TOTAL_ENERGY].sum().backward()
    return model_.mods["embedding"].embedding[1].weight.grad.clone()
for native in (True, False):
    gp, gm = gradE(m, pos + h * v, native), gradE(m, pos - h * v, native)
    q = -(gp - gm) / (2 * h)
    print("native" if native else "differentiable", "grad E(pos+hv)[3,5]", gp[3, 5].item(), "grad E(pos-hv)[3,5]", gm[3, 5].item(), "-> d(v.F)/dtheta[3,5]", q[3, 5].item(), "[10,20]", q[10, 20].item())
print("---- the same with Invariant eps = 1e-2 (nn/o3layer.py:39-44: sqrt(sum V^2 + eps^2) - eps has curvature 1 / eps at V = 0)")
for eps in (1e-2, 1e-5):
    ma, mb = _model(torch.float64, **SMALL).train(), _model(torch.float64, **SMALL).train()
    for mm in (ma, mb):
        for mod in mm.modules():
            if hasattr(mod, "invariant"): mod.invariant.eps = eps
    w = {keys.FORCES: 10.0}
    for disp in (2e-3, 2e-5, 2e-7):
        ma.zero_grad()
        train.train_step_directional(ma, dict(dev), tgt, _Frozen(ma.parameters()), w, order=4, displacement=disp)
        mb.zero_grad()
        train.train_step(mb, dict(dev), tgt, _Frozen(mb.parameters()), w)
        worst = max(((p.grad - q.grad).abs().max() / q.grad.abs().max()).item() for p, q in zip(ma.parameters(), mb.parameters()) if q.grad is not None)
        print(f"eps {eps:g} displacement {disp:g}: worst relative gradient error {worst:.2e}")
