"""Training pass on the GPU: parameter gradients and the double backward of a force / virial loss
(nn/basic.py:143-199 with create_graph=training, utils/trainer.py:290-308), against the fp64 oracle differentiated by
autograd w.r.t. its own state-dict entries, plus the data-parallel step (run/train.py:185-190) with two ranks on one card.

Tolerances: fp64 model vs fp64 oracle, relative to the largest entry of each gradient: 1e-8 (two different op orders of
the same arithmetic); fp32 training pass vs the fused inference kernels on energies / forces: the fp32 bounds of
tests/test_gpu_parity.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import xpainn_oracle as orc
from xequinet_amd import keys, train
from xequinet_amd.data import synthetic as syn
from xequinet_amd.nn import resolve_model

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL = dict(node_dim=128, node_irreps="128x0e + 64x1o + 32x2e", action_blocks=2, hidden_dim=64)


def _model(dtype, seed=0, **kw):
    torch.manual_seed(seed)
    model = resolve_model("xpainn", **kw)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith(("norm.weight", "affine_weight")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith(("bias", "affine_bias")):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return model.to(dtype).to(DEV)


def _batch(n_mol, seed, dtype, periodic=False):
    if periodic:
        pos, z, ptr, cell = syn.synth_water_box(3, seed=seed)              # 81 atoms, cubic box of 9.3 A
        ei, off = orc.radius_graph_pbc_oracle(pos, np.array([len(pos)]), [True, True, True], cell, 5.0)
        extra = {"cell": cell, "cell_offsets": off}
    else:
        pos, z, ptr = syn.synth_qm9_batch(n_mol, seed=seed)
        ei, extra = orc.radius_graph_canonical(pos, ptr, 5.0), {}
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    host = {"pos": torch.tensor(pos, dtype=torch.float64), "atomic_numbers": torch.tensor(z.astype(np.int64)),
            "edge_index": torch.tensor(ei), "batch": torch.tensor(batch), "ptr": torch.tensor(ptr)}
    for k, v in extra.items():
        host[k] = torch.tensor(v, dtype=torch.float64)
    dev = {k: (v.to(dtype) if v.is_floating_point() else v).to(DEV) for k, v in host.items()}
    return host, dev


def _targets(host, seed, virial):
    g = torch.Generator().manual_seed(seed)
    n_mol, n = host["ptr"].numel() - 1, host["pos"].shape[0]
    t = {keys.TOTAL_ENERGY: torch.randn(n_mol, generator=g, dtype=torch.float64),
         keys.FORCES: torch.randn(n, 3, generator=g, dtype=torch.float64), keys.BATCH_PTR: host["ptr"]}
    if virial:
        t[keys.VIRIAL] = torch.randn(n_mol, 3, 3, generator=g, dtype=torch.float64)
    return t


VARIANT = dict(node_dim=64, node_irreps="64x0e + 32x1o + 32x2e", action_blocks=2, hidden_dim=32, num_basis=12,
               rbf_kernel="gaussian", cutoff_fn="polynomial", layer_norm=False, activation="tanh")


@pytest.mark.parametrize("case", ["energy", "variant energy", "periodic energy", "energy (differentiable form)", "energy+forces",
                                  "periodic energy+forces+virial", "variant energy+forces", "energy+forces, 40 molecules",
                                  "expbern energy", "expbern energy+forces", "expnorm energy", "expnorm energy+forces"])
def test_parameter_gradients_match_the_oracle(case):
    """An energy-only loss takes the NATIVE training pass (fused kernels + xeq_message_param_grad, nn/fused.py); forces / virial in
    the loss need second order and take the differentiable form (nn/training.py)."""
    periodic = case.startswith("periodic")
    cfg = VARIANT if case.startswith("variant") else SMALL   # variant: gaussian basis (trainable mean / std), polynomial envelope, no layer norm, tanh
    if case.startswith("exp"):   # the exponential bases (nn/rbf.py:161-207; trainable _alpha / beta, mu): every training pass takes the differentiable form
        cfg = dict(SMALL, rbf_kernel=case.split()[0])
    weights = {keys.TOTAL_ENERGY: 1.0}
    if "forces" in case:
        weights[keys.FORCES] = 10.0
    if "virial" in case:
        weights[keys.VIRIAL] = 0.5
    model = _model(torch.float64, **cfg).train()
    model.native_training = "differentiable" not in case
    host, dev = _batch(40 if "40 molecules" in case else 6, 5, torch.float64, periodic)     # 40 molecules: ~720 atoms, 12 k edges
    tgt = _targets(host, 7, keys.VIRIAL in weights)
    data = dict(dev)
    result = model(data, keys.FORCES in weights, keys.VIRIAL in weights)
    from xequinet_amd.nn import training as tr
    assert bool(data[tr.PARAM_GRADS]) == (model.native_training and keys.FORCES not in weights and keys.VIRIAL not in weights
                                          and not case.startswith("exp"))
    loss, _ = train.weighted_loss(result, {k: v.to(DEV) for k, v in tgt.items()}, weights)
    loss.backward()

    sd = {k: v.detach().cpu().double().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    want = orc.XPaiNNOracle(sd, **cfg)(host, keys.FORCES in weights, keys.VIRIAL in weights, training=True)
    ref_loss, _ = train.weighted_loss(want, tgt, weights)
    names = [n for n, _ in model.named_parameters()]
    ref_grads = torch.autograd.grad(ref_loss, [sd[n] for n in names], allow_unused=True)
    assert abs(loss.item() - ref_loss.item()) <= 1e-9 * max(1.0, abs(ref_loss.item()))
    checked = 0
    for (name, p), g_ref in zip(model.named_parameters(), ref_grads):
        if g_ref is None:      # e.g. the head's last bias under a forces-only loss
            assert p.grad is None or p.grad.abs().max().item() == 0.0, name
            continue
        g_ref = g_ref.reshape(p.shape)
        err = (p.grad.cpu() - g_ref).abs().max().item()
        assert err <= 1e-8 * max(1e-6, g_ref.abs().max().item()), f"{name}: {err:.2e} of {g_ref.abs().max().item():.2e}"
        checked += 1
    assert checked >= len(names) - 1 and checked >= 38     # 38 parameter tensors without layer norms, 54 with


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_native_training_pass_with_atoms_that_have_no_neighbour(dtype):
    """A lone atom, a pair beyond the cutoff and an ordinary molecule in one batch: isolated nodes contribute no filter gradient and
    their node-side gradients come out right (f64: against the oracle; f32, the matrix-core forms: against the differentiable form)."""
    pos0, z0, ptr0 = syn.synth_qm9_batch(2, seed=21)
    pos = np.concatenate([pos0, [[30.0, 0.0, 0.0]], [[60.0, 0.0, 0.0], [60.0, 8.0, 0.0]]])
    z = np.concatenate([z0, [8], [1, 6]])
    ptr = np.concatenate([ptr0, [ptr0[-1] + 1, ptr0[-1] + 3]])
    ei = orc.radius_graph_canonical(pos, ptr, 5.0)
    batch = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    host = {"pos": torch.tensor(pos, dtype=torch.float64), "atomic_numbers": torch.tensor(z.astype(np.int64)),
            "edge_index": torch.tensor(ei), "batch": torch.tensor(batch), "ptr": torch.tensor(ptr)}
    dev = {k: (v.to(dtype) if v.is_floating_point() else v).to(DEV) for k, v in host.items()}
    tgt = _targets(host, 3, False)
    w = {keys.TOTAL_ENERGY: 1.0}
    model = _model(dtype, **SMALL).train()
    data = dict(dev)
    loss, _ = train.weighted_loss(model(data, False, False), {k: (v.to(dtype) if v.is_floating_point() else v).to(DEV) for k, v in tgt.items()}, w)
    loss.backward()
    from xequinet_amd.nn import training as tr
    assert data[tr.PARAM_GRADS]
    if dtype == torch.float64:
        sd = {k: v.detach().cpu().double().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        ref_loss, _ = train.weighted_loss(orc.XPaiNNOracle(sd, **SMALL)(host, False, False, training=True), tgt, w)
        names = [n for n, _ in model.named_parameters()]
        ref = dict(zip(names, torch.autograd.grad(ref_loss, [sd[n] for n in names], allow_unused=True)))
        tol = 1e-8
    else:
        other = _model(dtype, **SMALL).train()
        other.native_training = False
        ref_loss, _ = train.weighted_loss(other(dict(dev), False, False), {k: (v.to(dtype) if v.is_floating_point() else v).to(DEV) for k, v in tgt.items()}, w)
        ref_loss.backward()
        ref = {n: p.grad for n, p in other.named_parameters()}
        tol = 2e-4
    for n, p in model.named_parameters():
        g_ref = ref[n]
        if g_ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, n
            continue
        g_ref = g_ref.reshape(p.shape).double().cpu()
        err = (p.grad.double().cpu() - g_ref).abs().max().item()
        assert err <= tol * max(1e-6, g_ref.abs().max().item()), f"{n}: {err:.2e} of {g_ref.abs().max().item():.2e}"


def test_training_pass_gives_the_inference_numbers():
    """Same weights, same batch: train mode (differentiable form) against eval mode (the fused kernels), fp32."""
    model = _model(torch.float32, action_blocks=3)
    host, dev = _batch(48, 3, torch.float32)
    model.eval()
    with torch.enable_grad():
        want = model(dict(dev), True, False)
    model.train()
    got = model(dict(dev), True, False)
    assert got[keys.FORCES].requires_grad          # create_graph=training
    dE = (got[keys.TOTAL_ENERGY] - want[keys.TOTAL_ENERGY]).abs().max().item()
    dF = (got[keys.FORCES] - want[keys.FORCES]).abs()
    assert dE <= 1e-5 * want[keys.TOTAL_ENERGY].abs().max().item() + 1e-4
    assert dF.max().item() <= 1e-3 and torch.quantile(dF.flatten(), 0.99).item() <= 1e-4


@pytest.mark.parametrize("cfg_name", ["default", "variant"])
def test_native_training_pass_in_fp32_against_the_differentiable_form(cfg_name):
    """fp32, a batch large enough for the matrix-core message kernels (wq): the native pass and the ATen form of the same step give
    the same loss and parameter gradients to fp32 round-off (relative to each gradient's largest entry)."""
    cfg = dict(action_blocks=3) if cfg_name == "default" else VARIANT
    host, dev = _batch(64, 3, torch.float32)
    tgt = {k: (v.float() if v.is_floating_point() else v).to(DEV) for k, v in _targets(host, 1, False).items()}
    grads, losses = [], []
    for native in (True, False):
        model = _model(torch.float32, **cfg).train()
        model.native_training = native
        loss, _ = train.weighted_loss(model(dict(dev), False, False), tgt, {keys.TOTAL_ENERGY: 1.0})
        loss.backward()
        losses.append(loss.item())
        grads.append({n: p.grad.double().cpu() for n, p in model.named_parameters()})
    assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[1]))
    assert set(grads[0]) == set(grads[1])
    for n, g in grads[1].items():
        err = (grads[0][n] - g).abs().max().item()
        assert err <= 2e-4 * max(1e-6, g.abs().max().item()), f"{n}: {err:.2e} of {g.abs().max().item():.2e}"


def test_training_step_as_one_graph_follows_the_host_launched_steps():
    """train.GraphedTrainStep: neighbour list + native energy pass + Adam as ONE captured graph over capacity-sized arrays.  Three
    different batches (other atom, graph and edge counts) through the one capture; a twin model stepped by ``train_step`` on the
    same batches sees the same losses and ends with the same parameters (to fp32 rounding: the captured loss sums its graphs in
    another order)."""
    from xequinet_amd import runtime

    torch.manual_seed(0)
    batches = []
    for k, n_mol in enumerate((40, 33, 48)):
        host, dev = _batch(n_mol, 30 + k, torch.float32)
        tgt = _targets(host, 60 + k, False)
        batches.append((host, dev, tgt))
    cap = (max(b[0]["pos"].shape[0] for b in batches) + 8, max(b[0]["ptr"].numel() - 1 for b in batches),
           max(runtime.pair_capacity(b[0]["ptr"].numpy()) for b in batches))
    fast, slow = _model(torch.float32, **SMALL).train(), _model(torch.float32, **SMALL).train()
    slow.load_state_dict(fast.state_dict())
    opt_f = torch.optim.Adam(fast.parameters(), lr=1e-3, capturable=True)
    opt_s = torch.optim.Adam(slow.parameters(), lr=1e-3, capturable=True)
    step = train.GraphedTrainStep(fast, opt_f, cap)
    for host, dev, tgt in batches + batches[:1]:
        loss_f = step(dev["pos"], dev["atomic_numbers"], dev["ptr"], tgt[keys.TOTAL_ENERGY].float().to(DEV), batch=dev["batch"]).item()
        data = {k: v for k, v in dev.items()}
        from xequinet_amd.data import NeighborTransform, XequiBatch
        b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
        t = {keys.TOTAL_ENERGY: tgt[keys.TOTAL_ENERGY].float().to(DEV), keys.BATCH_PTR: dev["ptr"]}
        loss_s = train.train_step(slow, b.to_dict(), t, opt_s, {keys.TOTAL_ENERGY: 1.0})[0].item()
        assert abs(loss_f - loss_s) <= 2e-5 * max(1.0, abs(loss_s)), (loss_f, loss_s)
    assert step.captures == 1
    for (n, p), (_, q) in zip(fast.named_parameters(), slow.named_parameters()):
        scale = max(1e-3, q.abs().max().item())
        assert (p - q).abs().max().item() <= 2e-4 * scale, n


def test_captured_step_with_gradient_clipping_and_ema_follows_the_reference_loop():
    """utils/trainer.py:303-311 inside the capture: clip_grad_norm_ between the reverse pass and Adam, and the exponential moving average
    of the parameters behind it.  Against the host-launched loop in the reference's own form -- ``train_step(..., grad_clip=..)`` and a
    ``torch.optim.swa_utils.AveragedModel`` with avg_fn = decay avg + (1 - decay) p, whose first update is a copy -- over four steps
    on three different batches in fp64: losses, parameters and averaged parameters agree to 1e-9; the clip bites (the gradient norm of a
    random-weight model is far above the threshold), and the average differs from the parameters."""
    from torch.optim.swa_utils import AveragedModel

    from xequinet_amd import runtime
    from xequinet_amd.data import NeighborTransform, XequiBatch

    torch.manual_seed(0)
    dt, decay, clip = torch.float64, 0.9, 0.05
    batches = []
    for k, n_mol in enumerate((12, 9, 14)):
        host, dev = _batch(n_mol, 70 + k, dt)
        batches.append((host, dev, _targets(host, 80 + k, False)))
    cap = (max(b[0]["pos"].shape[0] for b in batches) + 8, max(b[0]["ptr"].numel() - 1 for b in batches),
           max(runtime.pair_capacity(b[0]["ptr"].numpy()) for b in batches))
    fast, slow = _model(dt, **SMALL).train(), _model(dt, **SMALL).train()
    slow.load_state_dict(fast.state_dict())
    opt_f = torch.optim.Adam(fast.parameters(), lr=1e-3, capturable=True)
    opt_s = torch.optim.Adam(slow.parameters(), lr=1e-3, capturable=True)
    ema_ref = AveragedModel(slow, device=DEV, avg_fn=lambda avg, p, num: decay * avg + (1 - decay) * p)     # trainer.py:218-227
    step = train.GraphedTrainStep(fast, opt_f, cap, grad_clip=clip, ema_decay=decay)
    for host, dev, tgt in batches + batches[:1]:
        loss_f = step(dev["pos"], dev["atomic_numbers"], dev["ptr"], tgt[keys.TOTAL_ENERGY].to(DEV), batch=dev["batch"]).item()
        b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
        t = {keys.TOTAL_ENERGY: tgt[keys.TOTAL_ENERGY].to(DEV), keys.BATCH_PTR: dev["ptr"]}
        loss_s = train.train_step(slow, b.to_dict(), t, opt_s, {keys.TOTAL_ENERGY: 1.0}, grad_clip=clip, ema_model=ema_ref)[0].item()
        assert abs(loss_f - loss_s) <= 1e-9 * max(1.0, abs(loss_s)), (loss_f, loss_s)
        gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in slow.parameters() if p.grad is not None)).item()
        assert gnorm <= clip * (1 + 1e-6)                                   # what train_step left behind is the clipped gradient
    assert step.captures == 1
    moved = 0.0
    for (n, p), q, a, (_, r) in zip(fast.named_parameters(), slow.parameters(), step.ema_parameters, ema_ref.module.named_parameters()):
        scale = max(1e-3, q.abs().max().item())
        assert (p - q).abs().max().item() <= 1e-9 * scale, n
        assert (a - r).abs().max().item() <= 1e-9 * scale, "ema of " + n
        moved = max(moved, (a - p).abs().max().item())
    assert moved > 1e-5
    twin = _model(dt, **SMALL)
    step.copy_ema_to(twin)
    assert all(torch.equal(a, b) for a, b in zip(twin.parameters(), step.ema_parameters))


def test_force_loss_training_step_as_one_graph_follows_the_host_launched_steps():
    """The same with forces in the loss: the twice-differentiable pass (kernel forms of nn/training.py) inside the capture, over the
    capacity-sized edge list -- padding atoms, empty graph slots and the edge slots behind the true count (stale pairs of earlier
    batches) must not reach the gradients.  In fp64, every replay against a host-launched evaluation on the exact edge list at EQUAL
    weights (the twin takes the captured model's weights before each step): 1e-8 of the largest entry of each gradient.  fp32 cannot
    make this comparison: the force-loss gradient of a random-weight model is ill-conditioned (fp32 against fp64 on the exact list:
    0.5 % - 25 % depending on the batch, scratch/dbg_f32_vs_f64.py), and a library GEMM that changes its kernel with the row count
    (257 rows against 256) moves fp32 gradients by as much."""
    from xequinet_amd import runtime
    from xequinet_amd.data import NeighborTransform, XequiBatch

    torch.manual_seed(0)
    dt = torch.float64
    batches = []
    for k, n_mol in enumerate((20, 14, 24)):
        host, dev = _batch(n_mol, 40 + k, dt)
        batches.append((host, dev, _targets(host, 70 + k, False)))
    cap = (max(b[0]["pos"].shape[0] for b in batches) + 8, max(b[0]["ptr"].numel() - 1 for b in batches),
           max(runtime.pair_capacity(b[0]["ptr"].numpy()) for b in batches))
    fast, slow = _model(dt, **SMALL).train(), _model(dt, **SMALL).train()
    opt_f = torch.optim.Adam(fast.parameters(), lr=1e-4, capturable=True)
    step = train.GraphedTrainStep(fast, opt_f, cap, energy_weight=1.0, forces_weight=5.0)
    weights = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 5.0}
    for host, dev, tgt in batches + batches[:1]:
        slow.load_state_dict(fast.state_dict())
        before = {n: p.detach().clone() for n, p in fast.named_parameters()}
        e_t, f_t = tgt[keys.TOTAL_ENERGY].to(DEV), tgt[keys.FORCES].to(DEV)
        loss_f = step(dev["pos"], dev["atomic_numbers"], dev["ptr"], e_t, batch=dev["batch"], target_forces=f_t).item()
        b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
        assert b.to_dict()["edge_index"].shape[1] < cap[2]                  # slots behind the list in the captured step
        slow.zero_grad(set_to_none=True)
        loss_s, _ = train.weighted_loss(slow(b.to_dict(), True, False), {keys.TOTAL_ENERGY: e_t, keys.FORCES: f_t, keys.BATCH_PTR: dev["ptr"]}, weights)
        loss_s.backward()
        assert abs(loss_f - loss_s.item()) <= 1e-10 * max(1.0, abs(loss_s.item())), (loss_f, loss_s.item())
        for (n, p), (_, q) in zip(fast.named_parameters(), slow.named_parameters()):
            if q.grad is None:
                continue
            scale = max(1e-6, q.grad.abs().max().item())
            assert (p.grad - q.grad).abs().max().item() <= 1e-8 * scale, n              # the replay's gradients live in the graph's pool
            assert not torch.equal(p.detach(), before[n]) or float(q.grad.abs().max()) == 0.0, n      # and the update was applied
    assert step.captures == 1


def test_evaluations_between_graphed_training_steps_see_the_current_weights():
    """Replays of train.GraphedTrainStep update the weights in place without bumping their version counters; the packed weight
    copies (MLP, U|V, linear, node-block programs; Python and C++ caches) key on those counters.  The step now bumps the pack
    epoch (include/xeq.h) after every replay, so an evaluation BETWEEN replays repacks: graphed steps, eval, more graphed steps,
    eval -- each evaluation equals that of a fresh twin holding the same parameters (advisor, round 3, high)."""
    from xequinet_amd import runtime
    from xequinet_amd.data import NeighborTransform, XequiBatch

    torch.manual_seed(1)
    host, dev = _batch(40, 31, torch.float32)
    tgt = _targets(host, 61, False)[keys.TOTAL_ENERGY].float().to(DEV)
    cap = (host["pos"].shape[0] + 8, host["ptr"].numel() - 1, runtime.pair_capacity(host["ptr"].numpy()))
    model = _model(torch.float32, **SMALL).train()
    opt = torch.optim.Adam(model.parameters(), lr=3e-3, capturable=True)
    step = train.GraphedTrainStep(model, opt, cap)
    b = NeighborTransform(5.0)(XequiBatch(dev["pos"], dev["atomic_numbers"], dev["ptr"]))
    seen = []
    for round_ in range(3):
        for _ in range(2):
            step(dev["pos"], dev["atomic_numbers"], dev["ptr"], tgt, batch=dev["batch"])
        model.eval()
        with torch.enable_grad():
            got = model(b.to_dict(), compute_forces=True)
        twin = _model(torch.float32, **SMALL).eval()
        twin.load_state_dict(model.state_dict())
        with torch.enable_grad():
            want = twin(b.to_dict(), compute_forces=True)
        assert torch.equal(got[keys.TOTAL_ENERGY].detach(), want[keys.TOTAL_ENERGY].detach()), round_
        assert torch.equal(got[keys.FORCES], want[keys.FORCES]), round_
        seen.append(got[keys.TOTAL_ENERGY].detach().clone())
        model.train()
    assert step.captures == 1 and not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def test_frozen_model_in_train_mode_stays_on_the_fused_path():
    model = _model(torch.float32, action_blocks=1).requires_grad_(False).train()
    _, dev = _batch(4, 2, torch.float32)
    with torch.enable_grad():
        out = model(dict(dev), True, False)
    assert not out[keys.FORCES].requires_grad


def test_train_step_lowers_the_loss():
    model = _model(torch.float32, **SMALL)
    host, dev = _batch(16, 9, torch.float32)
    tgt = {k: (v.float() if v.is_floating_point() else v).to(DEV) for k, v in _targets(host, 1, False).items()}
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    w = {keys.ENERGY_PER_ATOM: 1.0, keys.FORCES: 1.0}
    losses = [train.train_step(model, dev, tgt, opt, w, grad_clip=10.0)[0].item() for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


# ---- two ranks on one card: DistributedDataParallel over gloo with device tensors -------------------------------------------
def _ddp_worker(rank, world, port, out, with_forces=True):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _model(torch.float64, **SMALL).train()
    ddp = train.wrap_ddp(model, local_rank=0)
    host, dev = _batch(4, 100 + rank, torch.float64)             # every rank its own molecules
    tgt = {k: v.to(DEV) for k, v in _targets(host, 50 + rank, False).items()}
    w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 3.0} if with_forces else {keys.TOTAL_ENERGY: 1.0}
    loss, _ = train.weighted_loss(ddp(dict(dev), with_forces, False), tgt, w)
    loss.backward()
    out[rank] = {n: p.grad.cpu().numpy() for n, p in model.named_parameters()}
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("with_forces", [True, False])
def test_ddp_two_ranks_average_the_gradients(with_forces):
    """with_forces=False: an energy loss, i.e. the NATIVE training pass under DistributedDataParallel (its gradient hooks see the
    parameter gradients the fused block functions return)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_ddp_worker, args=(2, port, out, with_forces), nprocs=2, join=True)
    # the same two batches in this process, one after the other
    grads = []
    for rank in range(2):
        model = _model(torch.float64, **SMALL).train()
        host, dev = _batch(4, 100 + rank, torch.float64)
        tgt = {k: v.to(DEV) for k, v in _targets(host, 50 + rank, False).items()}
        w = {keys.TOTAL_ENERGY: 1.0, keys.FORCES: 3.0} if with_forces else {keys.TOTAL_ENERGY: 1.0}
        loss, _ = train.weighted_loss(model(dict(dev), with_forces, False), tgt, w)
        loss.backward()
        grads.append({n: p.grad.cpu().numpy() for n, p in model.named_parameters()})
    for n in grads[0]:
        mean = 0.5 * (grads[0][n] + grads[1][n])
        scale = max(1e-6, np.abs(mean).max())
        # the training pass aggregates with index_add (f64 atomics: the order of a node's sum differs from run to run)
        assert np.abs(out[0][n] - mean).max() <= 1e-8 * scale, n
        assert np.array_equal(out[0][n], out[1][n]), n            # every rank holds the same averaged gradient


F32_GRAD_TOL = 2e-5     # of the largest entry of a gradient; achieved on the GPU: 3.7e-7 (energy loss), 7.7e-7 (energy + forces): profiles/parity_r05.json


@pytest.mark.parametrize("case", ["energy", "energy+forces"])
def test_fp32_parameter_gradients_of_a_well_conditioned_model_against_the_fp64_oracle(case):
    """The fp32 INSTANTIATIONS of the training kernels against the ORACLE (round-4 review: they were only compared with this package's own
    tensor form): the native pass of an energy loss (k_message_param_grad_mc, k_wgrad, the fused blocks' reverse kernels) and the
    twice-differentiated pass of a force loss (k_message_bwd_sbq*, k_q_wgrad, tn::*<float, DUAL>, xeq::linear) in fp32 on the GPU, every
    parameter gradient against the fp64 oracle differentiated by autograd.  A random-weight model's force-loss gradient is
    ill-conditioned in fp32 at ONE spot (Invariant's sqrt(V^2 + eps^2) - eps on scalar channels that cross zero: 0.5-25 % of the largest
    entry), so the model here has the 0e block of every update_V bounded away from zero by construction -- the
    function is then well conditioned and the difference is the kernels' fp32 rounding."""
    from tests import parity_record

    weights = {keys.TOTAL_ENERGY: 1.0}
    if "forces" in case:
        weights[keys.FORCES] = 10.0
    model = _model(torch.float64, **SMALL)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("update_V.weight"):
                p[: 128 * 128] *= 0.02
            elif name.endswith("update_V.bias"):
                sign = torch.where(torch.rand(128, generator=g) < 0.5, -1.0, 1.0)
                p.copy_((sign * (1.0 + 0.5 * torch.rand(128, generator=g))).to(p))
    sd = {k: v.detach().cpu().double().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    model = model.float().train()
    host, dev = _batch(40, 5, torch.float32)
    tgt = _targets(host, 7, False)
    result = model(dict(dev), keys.FORCES in weights, False)
    loss, _ = train.weighted_loss(result, {k: (v.float() if v.is_floating_point() else v).to(DEV) for k, v in tgt.items()}, weights)
    loss.backward()
    want = orc.XPaiNNOracle(sd, **SMALL)(host, keys.FORCES in weights, False, training=True)
    ref_loss, _ = train.weighted_loss(want, tgt, weights)
    names = [n for n, _ in model.named_parameters()]
    ref_grads = torch.autograd.grad(ref_loss, [sd[n] for n in names], allow_unused=True)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(1.0, abs(ref_loss.item()))
    worst, worst_name, checked = 0.0, "", 0
    for (name, p), g_ref in zip(model.named_parameters(), ref_grads):
        if g_ref is None:
            continue
        g_ref = g_ref.reshape(p.shape)
        rel = (p.grad.double().cpu() - g_ref).abs().max().item() / max(1e-6, g_ref.abs().max().item())
        if rel > worst:
            worst, worst_name = rel, name
        checked += 1
    parity_record.add(dict(config=f"fp32 training pass ({case}), well-conditioned model, 40 molecules: parameter gradients vs fp64 oracle",
                           worst_relative_to_largest_entry=worst, parameter=worst_name, tensors=checked, bound=F32_GRAD_TOL))
    assert checked >= 50 and worst <= F32_GRAD_TOL, (worst_name, worst)
