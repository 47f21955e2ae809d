"""Probe: HIP-graph capture of the model while an RCCL process group (and its watchdog thread) is alive."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import NeighborTransform, XequiBatch
from xequinet_amd.nn import resolve_model
from xequinet_amd.runtime import GraphedModel
torch.manual_seed(0)
model = resolve_model("xpainn").eval().requires_grad_(False).to("cuda")
gm = GraphedModel(model)
pos, z, ptr = syn.synth_qm9_batch(64, seed=1)
for i in range(5):
    b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device="cuda"), torch.tensor(z, device="cuda"), torch.tensor(ptr, device="cuda")))
    out = gm(b.to_dict())
    dist.barrier()
torch.cuda.synchronize()
print("ok captures", gm.captures, float(out["energy"].sum()))
dist.destroy_process_group()
