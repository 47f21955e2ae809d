"""Round-5 kernels that collapse the small launches of a step (pytest -m gpu): the fused energy head with its saved reverse row, the
first message block's front half as one gather, degrees -> guarded row pointer in one launch, and one edge-gradient launch for all
message blocks.  Each against a float64 / integer restatement of the same arithmetic (nn/output.py:104-128, nn/xpainn.py:62-81,
128-139, data/transform.py:58-64, nn/basic.py:143-159)."""
import numpy as np
import pytest
import torch

from xequinet_amd.data import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_fused_head_energies_and_saved_reverse_row():
    from xequinet_amd.nn.fused import EnergyReadout
    from xequinet_amd.nn.output import EnergyOut

    torch.manual_seed(0)
    head = EnergyOut(node_dim=128, hidden_dim=64).to(DEV)
    n_per = [3, 17, 0, 29, 1, 40]
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(n_per)]), device=DEV)
    batch = torch.repeat_interleave(torch.arange(len(n_per), device=DEV), torch.tensor(n_per, device=DEV))
    n = int(ptr[-1])
    s = torch.randn(n, 128, device=DEV, requires_grad=True)
    assert EnergyReadout.supported(head.out_mlp, s)
    atomic, total = EnergyReadout.apply(s, head.out_mlp, batch, ptr)
    # float64 restatement
    import copy

    ref = copy.deepcopy(head.out_mlp).double()
    s64 = s.detach().double().requires_grad_()
    a64 = ref(s64).reshape(-1)
    t64 = torch.zeros(len(n_per), dtype=torch.float64, device=DEV).index_add(0, batch, a64)
    assert float((atomic.double() - a64).abs().max()) <= 3e-6 * max(1.0, float(a64.abs().max()))
    assert float((total.double() - t64).abs().max()) <= 3e-6 * max(1.0, float(t64.abs().max()))
    # reverse pass for gradients at both outputs, one of them a broadcast scalar (what the sum of the energies hands over)
    ga = torch.randn(n, device=DEV)
    gt = torch.full((len(n_per),), -1.0, device=DEV)
    (g,) = torch.autograd.grad([atomic, total], s, [ga, gt])
    (g64,) = torch.autograd.grad([a64, t64], s64, [ga.double(), gt.double()], retain_graph=True)
    assert float((g.double() - g64).abs().max()) <= 3e-6 * max(1.0, float(g64.abs().max()))
    (g1,) = torch.autograd.grad(EnergyReadout.apply(s, head.out_mlp, batch, ptr)[1].sum(), s)
    (g1_64,) = torch.autograd.grad(t64.sum(), s64)
    assert float((g1.double() - g1_64).abs().max()) <= 3e-6 * max(1.0, float(g1_64.abs().max()))
    # rows do not depend on the batch they sit in
    a_half, _ = EnergyReadout.apply(s[:20].detach(), head.out_mlp, batch[:20], torch.tensor([0, 3, 20], device=DEV))
    assert torch.equal(a_half, atomic[:20].detach())


def test_first_block_front_is_the_three_gathers():
    from xequinet_amd.nn import resolve_model
    from xequinet_amd.nn.fused import first_block_front

    torch.manual_seed(1)
    model = resolve_model("xpainn").eval().requires_grad_(False).to(DEV)
    emb, msg = model.mods["embedding"], model.mods["message_0"]
    for dt in (torch.int32, torch.int64):
        z = torch.tensor([1, 6, 8, 1, 9, 7, 6, 0, 86], device=DEV, dtype=dt)
        rows = emb._embedded_rows(z)
        s, h, xhat = first_block_front(msg, z, rows, z.shape[0])
        _, h_t, x0_t = msg._element_front
        assert torch.equal(s, rows.index_select(0, z.long()))
        assert torch.equal(h, h_t.index_select(0, z.long()))
        n, F = z.shape[0], 128
        assert torch.equal(xhat[: n * F].view(n, F), x0_t.index_select(0, z.long())) and float(xhat[n * F :].abs().max()) == 0.0


@pytest.mark.parametrize("n", [0, 1, 63, 1024, 1025, 18609, 36864])
def test_degrees_to_guarded_row_pointer_in_one_launch(n):
    from xequinet_amd import lib
    from xequinet_amd.lib import call, ptr, stream

    g = torch.Generator().manual_seed(n)
    deg = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(DEV)
    want = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(deg.long(), 0)]).to(torch.int32)
    total = int(want[-1])
    for cap, empty in ((-1, False), (total, False), (total - 1, total > 0)):
        rowptr = torch.full((n + 1,), -7, dtype=torch.int32, device=DEV)
        count = torch.full((1,), -7, dtype=torch.int32, device=DEV)
        run = torch.full((1,), 5, dtype=torch.int64, device=DEV)
        call("xeq_rowptr_from_degrees", ptr(deg), n, cap, ptr(rowptr), ptr(count), ptr(run), stream())
        assert int(count) == total and int(run) == total + 5
        assert torch.equal(rowptr, torch.zeros_like(want) if empty else want)
    assert lib.load().xeq_rowptr_from_degrees_max() >= 32768


def test_one_edge_gradient_launch_for_all_blocks_equals_one_per_block():
    """The model's forces with the deferral (one xeq_message_wq_edge_grad_sum) against the same model with one edge-gradient launch
    per block and autograd's sums: the same arithmetic up to the association of three float additions."""
    from xequinet_amd import ops
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.nn import resolve_model

    torch.manual_seed(0)
    model = resolve_model("xpainn").eval().requires_grad_(False).to(DEV)
    pos, z, ptr = syn.synth_qm9_batch(400, seed=5)     # above the wq kernels' minimum edge count
    b = NeighborTransform(5.0)(XequiBatch(torch.tensor(pos, dtype=torch.float32, device=DEV), torch.tensor(z, device=DEV), torch.tensor(ptr, device=DEV)))
    with torch.enable_grad():
        out = model(b.to_dict(), compute_forces=True)
    orig = ops.EdgeGradDeferral.register
    ops.EdgeGradDeferral.register = lambda self: False      # nobody registers: every block launches its own edge gradient
    try:
        with torch.enable_grad():
            ref = model(b.to_dict(), compute_forces=True)
    finally:
        ops.EdgeGradDeferral.register = orig
    assert torch.equal(out["energy"], ref["energy"])
    scale = float(ref["forces"].abs().max())
    assert float((out["forces"] - ref["forces"]).abs().max()) <= 2e-6 * max(1.0, scale)
    assert not torch.equal(out["forces"], torch.zeros_like(out["forces"]))
