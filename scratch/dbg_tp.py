import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_tp import _case
for mode, has_w, shared in (("uuu", True, True), ("uuw", True, False), ("uvw", True, True), ("uvu", True, False), ("uvv", True, True)):
    mod, in1, in2, out, ins, x, y, w, _ = _case(mode, has_w, shared, n=37)
    mod = mod.to("cuda")
    xs = [t.cuda().requires_grad_() for t in (x, y)]
    ws = None if w is None else w.cuda().requires_grad_()
    res = {}
    for fused in (False, True):
        mod.fused = fused
        o = mod(*xs) if (ws is None or mod.internal_weights) else mod(*xs, ws)
        g = torch.ones_like(o) * 0.37 + torch.arange(o.shape[1], device="cuda") * 0.01
        leaves = xs + ([ws] if (ws is not None and not mod.internal_weights) else ([mod.weight] if mod.weight_numel else []))
        res[fused] = [o.detach()] + list(torch.autograd.grad(o, leaves, g))
    print(mode, has_w, shared, "internal", mod.internal_weights, [f"{float((a - b).abs().max()):.2e}" for a, b in zip(res[True], res[False])])
sys.argv = ["x", "1500"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_tp.py")).read().split("rows = []")[0])
for name, build in (("selfmix", selfmix), ("cartesian", cartesian)):
    mod, i1, i2 = build(16)
    mod = mod.to("cuda").double()
    x = torch.randn(700, i1.dim, device="cuda", dtype=torch.float64, requires_grad=True); y = torch.randn(700, i2.dim, device="cuda", dtype=torch.float64, requires_grad=True)
    w = None if mod.internal_weights else torch.randn(700, mod.weight_numel, device="cuda", dtype=torch.float64, requires_grad=True)
    res = {}
    for fused in (False, True):
        mod.fused = fused
        o = mod(x, y) if w is None else mod(x, y, w)
        g = torch.sin(torch.arange(o.numel(), device="cuda", dtype=torch.float64)).reshape(o.shape)
        res[fused] = [o.detach()] + list(torch.autograd.grad(o, [x, y] + ([w] if w is not None else [mod.weight]), g))
    print(name, [f"{float((a - b).abs().max() / (b.abs().max() + 1e-300)):.2e}" for a, b in zip(res[True], res[False])])
