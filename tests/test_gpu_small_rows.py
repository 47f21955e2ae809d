"""The few-row forms of the node-side products (csrc/xeq_linear.hip k_linear_s, ...): 16 x 16 exact-f32 tiles for systems that do not
fill the chip (MD-sized: interface/jit_model.py:148-216).  An exact-f32 matrix instruction is a chain of fused multiply-adds in k
order whatever its tile shape, so the forms must agree BIT FOR BIT with the 32-row forms: the row count picks a form, never a result
(sharded == unsharded across the size threshold)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


class _forms:
    """Run with the few-row forms forced off (XEQ_SMALL_ROWS=0) or on for every row count (a huge limit); read per call by the library."""

    def __init__(self, limit):
        self.limit = str(limit)

    def __enter__(self):
        self.old = os.environ.get("XEQ_SMALL_ROWS")
        os.environ["XEQ_SMALL_ROWS"] = self.limit

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("XEQ_SMALL_ROWS", None)
        else:
            os.environ["XEQ_SMALL_ROWS"] = self.old


def _both(fn):
    with _forms(0):
        a = fn()
    with _forms(1 << 40):
        b = fn()
    torch.cuda.synchronize()
    return a, b


def test_exact_f32_matrix_instructions_round_like_a_sequential_fmaf_chain():
    """The hardware property every few-row form rests on (xeq_mfma_order_probe): v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 fed the
    same k sequence give the same bits, and both equal a sequential fmaf chain per output element -- over operands of mixed magnitudes."""
    from xequinet_amd.lib import call, ptr, stream

    g = torch.Generator(device="cuda").manual_seed(0)
    diff = torch.zeros(2, dtype=torch.int32, device=_dev())
    for trial in range(40):
        scale = (1.0, 100.0, 1e-3, 1e4)[trial % 4]
        a = (torch.rand(32, 224, device=_dev(), generator=g) - 0.5) * scale
        b = torch.rand(224, 32, device=_dev(), generator=g) - 0.5
        if trial % 5 == 0:
            a = a * torch.exp(8.0 * (torch.rand(32, 224, device=_dev(), generator=g) - 0.5))     # eight decades inside one row
        call("xeq_mfma_order_probe", ptr(a.contiguous()), ptr(b.contiguous()), ptr(diff), stream())
    assert diff.tolist() == [0, 0], f"outputs differing: between the instruction shapes {int(diff[0])}, from the fmaf chain {int(diff[1])}"


@pytest.mark.parametrize("n", [1, 15, 16, 21, 33, 192, 1000])
@pytest.mark.parametrize("k_in,n_out,bias", [(224, 128, False), (128, 224, False), (56, 128, True), (128, 64, True), (256, 256, True)])
def test_linear_few_row_form_changes_no_bit(n, k_in, n_out, bias):
    from xequinet_amd.nn import fused

    torch.manual_seed(n + k_in)
    lin = torch.nn.Linear(k_in, n_out, bias=bias).to(_dev())
    x = torch.randn(n, k_in, device=_dev()) * 3.0
    pack = fused._linear_pack(lin, lin.weight, lin.bias, False)
    assert pack is not None
    for act in (0, 1):
        (y0, p0), (y1, p1) = _both(lambda: fused._linear(x, pack, k_in, n_out, bias, act=act, want_pre=True))
        assert torch.equal(y0, y1) and torch.equal(p0, p1)
        if act == 0:   # and the result is the product
            ref = torch.nn.functional.linear(x.double(), lin.weight.double(), None if lin.bias is None else lin.bias.double())
            assert (y1.double() - ref).abs().max() < 1e-4 * (1.0 + ref.abs().max())
    if n >= 16:   # gathered rows (the embedding's table lookup)
        idx = torch.randint(0, n, (n,), device=_dev(), dtype=torch.int32)
        (y0, _), (y1, _) = _both(lambda: fused._linear(x, pack, k_in, n_out, bias, row_index=idx))
        assert torch.equal(y0, y1)


@pytest.mark.parametrize("n", [1, 16, 21, 33, 192, 1000])
@pytest.mark.parametrize("k1,n2", [(128, 576), (352, 480), (128, 32), (32, 128)])
def test_mlp2_few_row_form_changes_no_bit(n, k1, n2):
    from xequinet_amd.nn import fused

    torch.manual_seed(n + k1)
    seq = torch.nn.Sequential(torch.nn.Linear(k1, 128), torch.nn.SiLU(), torch.nn.Linear(128, n2)).to(_dev()).requires_grad_(False)
    with torch.no_grad():
        seq[0].bias.normal_()
        seq[2].bias.normal_()
    wide = torch.randn(n, k1 + 8, device=_dev()) * 2.0
    x = wide[:, :k1]                                     # strided rows, as the update MLP reads them
    g = torch.randn(n, n2, device=_dev())
    (pre0, y0), (pre1, y1) = _both(lambda: fused._mlp_fwd(seq, x))
    assert torch.equal(pre0, pre1) and torch.equal(y0, y1)
    gx0, gx1 = _both(lambda: fused._mlp_bwd(seq, g, pre0))
    assert torch.equal(gx0, gx1)
    ref = torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(x.double(), seq[0].weight.double(), seq[0].bias.double())),
                                     seq[2].weight.double(), seq[2].bias.double())
    assert (y1.double() - ref).abs().max() < 1e-4 * (1.0 + ref.abs().max())


@pytest.mark.parametrize("n", [1, 16, 21, 33, 192, 1000])
@pytest.mark.parametrize("irreps,layer_norm", [("128x0e + 64x1o + 32x2e", True), ("128x0e + 64x1o + 32x2e", False)])
def test_update_block_few_row_forms_change_no_bit(n, irreps, layer_norm):
    """XPainnUpdate.forward (nn/xpainn.py:206-231) and its input gradients: norms -> U, V -> v, p (k_update_uv_fwd_s), the update MLP,
    dot_lin, and the reverse of all of them (k_update_uv_bwd_s, k_mlp2_s<true>, k_linear_s) against the 32-row forms."""
    from xequinet_amd.nn import fused
    from xequinet_amd.nn.xpainn import XPainnUpdate

    torch.manual_seed(n)
    blk = XPainnUpdate(node_dim=128, node_irreps=irreps, layer_norm=layer_norm).to(_dev()).eval().requires_grad_(False)
    with torch.no_grad():
        for prm in blk.parameters():
            if prm.dim() == 1:
                prm.add_(0.3 * torch.randn_like(prm))
    D = blk.node_irreps.dim
    s0, x0 = torch.randn(n, 128, device=_dev()), torch.randn(n, D, device=_dev())
    gs, gx = torch.randn(n, 128, device=_dev()), torch.randn(n, D, device=_dev())

    def run():
        s, x = s0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        with torch.enable_grad():
            so, xo = fused.UpdateBlock.apply(s, x, blk)
            g = torch.autograd.grad([so, xo], [s, x], [gs, gx])
        return so.detach(), xo.detach(), g[0], g[1]

    a, b = _both(run)
    assert getattr(blk, "_uv_frag", None) is not None, "the matrix-core front did not run"
    for name, u, v in zip(("s_out", "x_out", "grad_s", "grad_x"), a, b):
        assert torch.equal(u, v), f"{name}: {(u - v).abs().max().item():.2e}"


@pytest.mark.parametrize("system", ["aspirin", "qm9_8", "qm9_64", "qm9_230"])
def test_whole_evaluation_does_not_depend_on_the_few_row_forms(system):
    """Energies and forces of MD-sized systems (interface/jit_model.py:148-216) with every few-row form on against all of them off: bit for bit, eager and through the captured whole step.
    qm9_230 (4.1 k atoms) lies across the forms' row limit: a batch and its shards take different forms and still agree."""
    from tests.test_gpu_parity import _build, _t
    from xequinet_amd import runtime
    from xequinet_amd.data import NeighborTransform, XequiBatch
    from xequinet_amd.data import synthetic as syn

    model, _ = _build(torch.float32)
    if system == "aspirin":
        pos, z, _ = syn.synth_aspirin()
        ptr = torch.tensor([0, len(z)]).numpy()
    else:
        pos, z, ptr = syn.synth_qm9_batch(int(system.split("_")[1]), seed=5)

    def eager():
        b = NeighborTransform(5.0)(XequiBatch(_t(pos, torch.float32), _t(z), _t(ptr)))
        with torch.enable_grad():
            out = model(b.to_dict(), compute_forces=True)
        return out["energy"].detach().clone(), out["forces"].detach().clone()

    (e0, f0), (e1, f1) = _both(eager)
    assert torch.equal(e0, e1) and torch.equal(f0, f1)

    def graphed():
        step = runtime.GraphedStep(model, (len(pos) + 3, len(ptr) - 1, runtime.pair_capacity(ptr)))
        out = step(_t(pos, torch.float32), _t(z), _t(ptr))
        return out["energy"].clone(), out["forces"].clone()

    (e2, f2), (e3, f3) = _both(graphed)
    assert torch.equal(e2, e3) and torch.equal(f2, f3) and torch.equal(e2, e0) and torch.equal(f2, f0)
    if system == "qm9_230":   # the first molecules of the batch alone (below the limit) against their rows in the whole batch (above it)
        g = 50
        n = int(ptr[g])
        b = NeighborTransform(5.0)(XequiBatch(_t(pos[:n], torch.float32), _t(z[:n]), _t(ptr[: g + 1])))
        with torch.enable_grad():
            part = model(b.to_dict(), compute_forces=True)
        assert len(pos) > 3584 >= n
        assert torch.equal(part["energy"].detach(), e1[:g]) and torch.equal(part["forces"].detach(), f1[:n])
