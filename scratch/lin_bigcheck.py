"""The linear test kernel (no spills, ring only) at a size that puts two workgroups on every CU: sporadic errors?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.lib import call, ptr, stream
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
torch.manual_seed(0)
for form, n_ot in ((1, 18), (2, 18), (0, 4)):
    x = torch.randn(n, 128, device=dev); w = torch.randn(32 * n_ot, 128, device=dev)
    scratch = torch.empty((8 * n_ot + 32) * 3072, dtype=torch.uint8, device=dev)
    ref = None
    for rep in range(6):
        y = torch.full((n, 32 * n_ot), float("nan"), device=dev)
        call("xeq_node_block_linear_test", ptr(x), n, ptr(w), n_ot, form, ptr(scratch), ptr(y), stream())
        torch.cuda.synchronize()
        if ref is None:
            ref = (x.double() @ w.double().T)
            err = float((y.double() - ref).abs().max())
            print(f"form {form}: max |y - f64| {err:.2e}")
            first = y.clone()
        else:
            d = (y != first).any(1)
            print(f"form {form} rep {rep}: rows that differ from the first run: {int(d.sum())}", torch.nonzero(d).flatten()[:8].tolist())
