"""Outputs of the scalar-broadcast message kernels (forward + reverse) on fixed inputs -> argv[1] (.pt); run once per library
(XEQ_LIB_PATH) and compare the files: python scratch/sb_ab.py out.pt | python scratch/sb_ab.py --compare a.pt b.pt"""
import os, sys, math, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for k in a:
        same = torch.equal(a[k], b[k])
        bad += not same
        print(k, "equal" if same else f"DIFFERENT max {(a[k] - b[k]).abs().max().item():.3e}")
    sys.exit(1 if bad else 0)
os.environ["XEQ_MESSAGE_IMPL"] = "sb"
from oracle import xpainn_oracle as orc
from xequinet_amd import ops
from xequinet_amd.data import synthetic as syn
dev = torch.device("cuda", 0)
out = {}
for n_mol in (1, 8, 64):
    rng = np.random.default_rng(3)
    pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=11)
    ei = orc.radius_graph_canonical(pos.astype(np.float32), ptr, 5.0)
    N = len(pos)
    node_dim, mul, B = 128, (128, 64, 32), 20
    C, D = sum(mul), mul[0] + 3 * mul[1] + 5 * mul[2]
    H = node_dim + 2 * C
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    h, xhat, s, x = (f32(rng.normal(size=sh)) for sh in ((N, H), (N, D), (N, node_dim), (N, D)))
    W, b = f32(rng.normal(size=(H, B)) / math.sqrt(B)), f32(rng.normal(size=(H,)))
    p0 = f32(math.pi * np.arange(1, B + 1) / 5.0).view(1, -1)
    vec = f32(pos[ei[0]] - pos[ei[1]])
    gs, gx = f32(rng.normal(size=(N, node_dim))), f32(rng.normal(size=(N, D)))
    cfg = ("bessel", "cosine", B, 5.0, node_dim, mul)
    graph = ops.EdgeGraph(torch.tensor(ei, device=dev), N, ptr=torch.tensor(ptr, device=dev), symmetric=True)
    s_out, x_out, saved, impl = ops.message_forward(h, xhat, vec, s, x, W, b, p0, None, graph, cfg, want_backward=True)
    assert impl == "sb"
    g = ops.message_backward(saved, graph, cfg, impl, gs, gx, node_grads=True)
    for name, t in zip(("s_out", "x_out", "g_h", "g_xhat", "g_vec"), (s_out, x_out, g[0], g[1], g[2])):
        out[f"{n_mol}/{name}"] = t.cpu()
torch.save(out, sys.argv[1])
print("saved", len(out), "tensors")
