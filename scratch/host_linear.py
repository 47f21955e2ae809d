import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.interface import scripted
scripted.load_torch_library()
from xequinet_amd.nn import training_ops as tops
dev = "cuda"
x = torch.randn(512, 128, device=dev, requires_grad=True); W = torch.randn(256, 128, device=dev, requires_grad=True); b = torch.randn(256, device=dev, requires_grad=True)
t = torch.randn(512, 256, device=dev)
def run(fn, reps=300):
    for _ in range(20):
        y = fn(x, W, b); (gx,) = torch.autograd.grad((y * t).sum(), x, create_graph=True); torch.autograd.grad((gx * gx).sum(), (x, W, b), allow_unused=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        y = fn(x, W, b); (gx,) = torch.autograd.grad((y * t).sum(), x, create_graph=True); torch.autograd.grad((gx * gx).sum(), (x, W, b), allow_unused=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("F.linear        %.0f us per fwd + grad + gradgrad" % run(torch.nn.functional.linear))
print("xeq::linear     %.0f us" % run(torch.ops.xeq.linear))
print("python LinearFn %.0f us" % run(tops.LinearFn.apply))
