"""Probe: timing events recorded inside a captured HIP graph (torch.cuda.Event(external=True))."""
import torch
x = torch.randn(4096, 4096, device="cuda")
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): y = x @ x
torch.cuda.current_stream().wait_stream(side)
e1 = torch.cuda.Event(enable_timing=True, external=True)
e2 = torch.cuda.Event(enable_timing=True, external=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    z = x + 1
    e1.record()
    y = x @ x
    e2.record()
    w = y * 2
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "elapsed ms", e1.elapsed_time(e2))
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); y = x @ x; b.record(); torch.cuda.synchronize(); print("eager ms", a.elapsed_time(b))
