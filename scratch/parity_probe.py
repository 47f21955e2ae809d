"""HIP fp32 forces against the fp64 oracle, next to the fp32 oracle's own error, on the sampled molecules of the full-size
configurations (what tests/test_gpu_fullsize.py asserts): python scratch/parity_probe.py [workload ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _build, f32_force_bounds
from tests.test_gpu_fullsize import _hip_eval, _oracle_subset
from xequinet_amd.data import synthetic as syn
model, oracle = _build(torch.float32)
if os.environ.get("XEQ_PROBE_LIBGEMM"):      # development: node-side contractions through the library GEMMs instead of the MFMA kernels
    from xequinet_amd.nn import fused
    which = os.environ["XEQ_PROBE_LIBGEMM"]
    if "mlp" in which: fused._mlp_packs = lambda seq: None
    if "uv" in which: fused._packed_uv_frag = lambda module: None
    print("library GEMMs for:", which)
print("lib", os.environ.get("XEQ_LIB_PATH", "in-tree"), "impl", os.environ.get("XEQ_MESSAGE_IMPL", "auto"))
for name in sys.argv[1:] or ["qm9_1024", "md17_4096", "qm9_8192"]:
    pos, z, ptr, _ = syn.make_workload(name, seed=1234)
    n_mol = len(ptr) - 1
    E, F, e_hip, _ = _hip_eval(model, pos, z, ptr)
    mols = np.sort(np.random.default_rng(7).choice(n_mol, size=min(int(os.environ.get('XEQ_PROBE_SAMPLE', '64')), n_mol), replace=False))
    idx, Eref, Fref, e_sub, ref_in = _oracle_subset(oracle, pos, z, ptr, mols)
    dF = np.abs(F[idx] - Fref)
    b_max, b_p99, e_max, e_p99 = f32_force_bounds(oracle, ref_in, Fref)
    in32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ref_in.items()}
    from tests.test_gpu_parity import f32_twin
    e32 = np.abs(f32_twin(oracle)(in32, compute_forces=True)["forces"].double().numpy() - Fref)
    q = lambda a: " ".join(f"{np.quantile(a, x):.2e}" for x in (0.5, 0.9, 0.99, 0.999)) + f" max {a.max():.2e} rms {np.sqrt((a**2).mean()):.2e}"
    print(f"{name} [{len(mols)} mol] p50 p90 p99 p999: HIP {q(dF)} | oracle32 {q(e32)}")
    print(f"{name}: HIP max {dF.max():.2e} p99 {np.quantile(dF, .99):.2e} p999 {np.quantile(dF, .999):.2e} rms {np.sqrt((dF**2).mean()):.2e} | "
          f"oracle32 max {e_max:.2e} p99 {e_p99:.2e} | bounds {b_max:.2e} {b_p99:.2e} | {'ok' if dF.max() <= b_max and np.quantile(dF, .99) <= b_p99 else 'FAIL'}", flush=True)
