"""Does the PHASE between the two steps in flight matter?  The second context's stream starts `delay` microseconds late (one spin kernel in
front of its first replay); steps take equal time, so the offset persists.  python scratch/in_flight_phase.py [delays in us ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xequinet_amd.nn import resolve_model
from xequinet_amd.data import synthetic as syn
from xequinet_amd import runtime

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to(torch.float32).to(dev)
pos, z, ptr, _ = syn.make_workload("qm9_1024", seed=1234)
t = lambda a, dt=None: torch.as_tensor(a, device=dev).to(dt) if dt is not None else torch.as_tensor(a, device=dev)
a = (t(pos, torch.float32), t(z), t(ptr))
cap = (len(pos) + 64, len(ptr) - 1, runtime.pair_capacity(ptr))
fl = runtime.GraphedStepsInFlight(model, cap, depth=2)
for _ in range(4):
    fl.submit(*a)
torch.cuda.synchronize()
CYC_PER_US = 100      # torch.cuda._sleep counts in ~10 ns ticks of the 100 MHz clock on this stack (calibrated below)
t0 = time.perf_counter(); torch.cuda._sleep(1_000_000); torch.cuda.synchronize(); per = (time.perf_counter() - t0) * 1e6 / 1_000_000
print(f"_sleep: {per * 1000:.2f} ns per count")
for delay in [float(x) for x in sys.argv[1:]] or [0, 250, 500, 1000, 1500]:
    n = 80
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        if k == 1 and delay > 0:
            with torch.cuda.stream(fl._streams[1]):
                torch.cuda._sleep(int(delay / per))
        fl.submit(*a)
    torch.cuda.synchronize()
    ms = ((time.perf_counter() - t0) * 1e3 - delay * 1e-3 * 0.5) / n
    print(f"second stream {delay:6.0f} us late: {ms:.4f} ms per step", flush=True)
