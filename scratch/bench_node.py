"""Node-side stages alone (one XPainnUpdate block forward + reverse) on QM9-1024-sized random features; run under
rocprofv3 --kernel-trace --stats to read the per-kernel averages."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import resolve_model
from xequinet_amd.nn.fused import UpdateBlock
dev = "cuda"
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(dev)
mod = model.mods["update_0"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 18609
s = torch.randn(N, 128, device=dev); x = torch.randn(N, 480, device=dev)
gs = torch.randn(N, 128, device=dev); gx = torch.randn(N, 480, device=dev)
for _ in range(20):
    ss, xx = s.clone().requires_grad_(), x.clone().requires_grad_()
    so, xo = UpdateBlock.apply(ss, xx, mod)
    torch.autograd.backward([so, xo], [gs, gx])
torch.cuda.synchronize()
print("done")
