import os, sys, ctypes, subprocess, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import xpainn_oracle as orc
from xequinet_amd.data import synthetic as syn
from xequinet_amd.data import NeighborTransform, XequiBatch
so = "/tmp/gather_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "scratch/gather_probe.hip"), "-o", so])
hip = ctypes.CDLL("libamdhip64.so"); mod = ctypes.c_void_p(); 
# simpler: use torch's cpp? fall back to hipModuleLoad on the code object inside the .so is awkward -> launch through hipLaunchKernel via symbol address
lib = ctypes.CDLL(so)
pos, z, ptr = syn.synth_qm9_batch(1024, seed=1234)
b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32), torch.tensor(z), torch.tensor(ptr)).to("cuda"))
g = getattr(b, "_xeq_edge_graph"); N, E = g.n_nodes, g.n_edges
W = 1024
rows = torch.randn(N, W, device="cuda"); out = torch.zeros(N, 256, device="cuda")
nbr = b.edge_index[1].contiguous()
hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, ctypes.c_uint*3, ctypes.c_uint*3, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
class dim3(ctypes.Structure): _fields_=[("x",ctypes.c_uint),("y",ctypes.c_uint),("z",ctypes.c_uint)]
hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, dim3, dim3, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
ebuf = torch.randn(E, 32, device="cuda"); wts = torch.randn(768, 20, device="cuda")
def launch(name, grid):
    fn = ctypes.cast(getattr(lib, name), ctypes.c_void_p)
    a = [ctypes.c_void_p(g.c_rowptr.data_ptr()), ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(rows.data_ptr()), ctypes.c_int64(N), ctypes.c_int(W), ctypes.c_void_p(out.data_ptr())]
    if name == "probe_fma": a += [ctypes.c_void_p(ebuf.data_ptr()), ctypes.c_void_p(wts.data_ptr())]
    arr = (ctypes.c_void_p * len(a))(*[ctypes.cast(ctypes.pointer(x), ctypes.c_void_p) for x in a])
    rc = hip.hipLaunchKernel(fn, dim3(grid,1,1), dim3(256,1,1), arr, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
for name in ("probe_chan", "probe_fma"):
    for grid in (1024, 2048):
        launch(name, grid); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): launch(name, grid)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        print(f"{name} grid {grid}: {us:7.1f} us  -> {E * W * 4 / us / 1e6:6.2f} TB/s gathered")
ref = torch.zeros(N, W, device="cuda").index_add_(0, b.edge_index[0], rows[nbr]); ref = ref.view(N,4,256).sum(1)
launch("probe_wave", 1024); torch.cuda.synchronize(); print("check", float((out-ref).abs().max()))
