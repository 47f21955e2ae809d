"""Where the periodic neighbour search of water-512 spends its time (host-synchronised pieces)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import radius_graph as rg
from xequinet_amd import ops
dev = "cuda"
pos, z, ptr, cell = syn.synth_water_box(8, seed=5)
p = torch.tensor(pos, dtype=torch.float32, device=dev); c = torch.tensor(cell, dtype=torch.float32, device=dev)
pbc = torch.tensor([[True, True, True]], device=dev); npg = torch.tensor([len(pos)], device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("radius_graph_pbc total: %.0f us" % t(lambda: rg.radius_graph_pbc(p, npg, pbc, c, 5.0)))
pbc_ = [True, True, True]
print("  pbc.cpu(): %.0f us" % t(lambda: pbc.detach().cpu()))
print("  _image_counts: %.0f us" % t(lambda: rg._image_counts(c, pbc_, 5.0, with_prune=True)))
print("  wrap_positions: %.0f us" % t(lambda: rg.wrap_positions(p, c, npg, pbc_)))
cpa = c.repeat_interleave(npg, dim=0)
print("    linalg.inv [N,3,3]: %.0f us" % t(lambda: torch.linalg.inv(cpa)))
print("    linalg.inv [1,3,3]: %.0f us" % t(lambda: torch.linalg.inv(c)))
max_rep, prune = rg._image_counts(c, pbc_, 5.0, with_prune=True)
cpd = [torch.arange(-r, r + 1, device=dev, dtype=torch.float32) for r in max_rep]
co = torch.cartesian_prod(*cpd).reshape(-1, 3)
po = torch.bmm(co.view(1, -1, 3).expand(1, -1, -1).contiguous(), c)
pw, sh = rg.wrap_positions(p, c, npg, pbc_)
pt = torch.tensor([0, len(pos)], device=dev)
print("  images (arange, cartesian_prod, bmm): %.0f us" % t(lambda: torch.bmm(torch.cartesian_prod(*[torch.arange(-r, r + 1, device=dev, dtype=torch.float32) for r in max_rep]).reshape(-1, 3).view(1, -1, 3).contiguous(), c)))
print("  raw search (pruned): %.0f us" % t(lambda: ops.radius_graph_pbc_raw(pw, pt, po, co, sh, 5.0, prune=prune)))
print("  raw search (cell list): %.0f us" % t(lambda: ops.radius_graph_pbc_raw(pw, pt, po, co, sh, 5.0, prune=rg._with_bins(prune, pbc_))))
