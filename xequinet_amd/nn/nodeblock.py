"""Host side of the fused node-block kernels (csrc/xeq_nodeblock.hip): XPainnUpdate.forward (nn/xpainn.py:206-231 of the
reference) together with the front half of the next XPainnMessage.forward (nn/xpainn.py:128-139) as ONE launch, and the same chain
backwards for the force evaluation (nn/basic.py:143-159).

The packed weight programs are cached per (update module, next message module) and rebuilt when any weight moves
(``lib.pack_epoch`` covers in-place updates that do not bump tensor versions, e.g. a replayed captured optimizer step).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import lib
from ..lib import call, mul3, ptr, stream

TILE_BYTES = 3072
pack_epoch = lib.pack_epoch   # every pack cache keys on it next to the tensors' version counters (include/xeq.h)


def supported(update, message=None) -> bool:
    """Whether the fused node-block kernels take this update block (and, if given, the next message block's front half)."""
    from .fused import _packed_uv

    if not isinstance(update.update_mlp[1], torch.nn.SiLU) or isinstance(update.norm, torch.nn.Identity):
        return False
    w = update.update_mlp[0].weight
    if w.dtype != torch.float32 or not w.is_cuda:
        return False
    if not lib.load().xeq_node_block_supported(lib.XEQ_F32, update.node_dim, mul3(update.node_irreps.mul3())):
        return False
    if update.update_mlp[0].bias is None or update.update_mlp[2].bias is None or update.dot_lin.bias is not None:
        return False
    if message is not None:
        if (not isinstance(message.scalar_mlp[1], torch.nn.SiLU) or isinstance(message.norm, torch.nn.Identity)
                or message.node_dim != update.node_dim or tuple(message._mul) != tuple(update.node_irreps.mul3())
                or message.scalar_mlp[0].bias is None or message.scalar_mlp[2].bias is None):
            return False
    return True


def _versions(*tensors):
    return tuple((t._version, t.data_ptr()) for t in tensors if t is not None)


def packed_fwd(update, message=None) -> torch.Tensor:
    """The forward weight program of (update block, next message block or None), cached on the update module."""
    from .fused import _packed_uv

    m = update.update_mlp
    ws = [m[0].weight, update.update_U.weight, update.update_V.weight, update.dot_lin.weight, m[2].weight]
    if message is not None:
        ws += [message.scalar_mlp[0].weight, message.scalar_mlp[2].weight]
    key = (_versions(*ws), pack_epoch(), id(message))
    cache = getattr(update, "_xeq_nb_fwd", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        uv, _ = _packed_uv(update)        # [W_U | W_V] / sqrt(mul) per l, [mul, 2 mul]
        tiles = int(lib.load().xeq_node_block_fwd_tiles(int(message is not None)))
        out = torch.empty(tiles * TILE_BYTES, dtype=torch.uint8, device=m[0].weight.device)
        w1n = message.scalar_mlp[0].weight.detach().contiguous() if message is not None else None
        w2n = message.scalar_mlp[2].weight.detach().contiguous() if message is not None else None
        keep = [m[0].weight.detach().contiguous(), update.dot_lin.weight.detach().contiguous(), m[2].weight.detach().contiguous()]
        call("xeq_node_block_pack_fwd", ptr(keep[0]), ptr(uv[0]), ptr(uv[1]), ptr(uv[2]), ptr(keep[1]), ptr(keep[2]), ptr(w1n), ptr(w2n),
             ptr(out), stream())
    update._xeq_nb_fwd = (key, out)
    return out


def _uv_bias(update) -> Optional[torch.Tensor]:
    bu, bv = update.update_U.bias, update.update_V.bias
    if bu is None or bu.numel() == 0:
        return None
    key = (_versions(bu, bv), pack_epoch())
    cache = getattr(update, "_xeq_nb_buv", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        b = torch.cat([bu.detach(), bv.detach()]).contiguous()
    update._xeq_nb_buv = (key, b)
    return b


def node_block_fwd(s: torch.Tensor, x: torch.Tensor, update, message=None, want_x: bool = True) -> dict:
    """One launch: the update block on (s, x) and, with ``message``, the next message block's norms + scalar_mlp on the result.
    Returns the tensors the reverse pass and the message kernel read (layouts of include/xeq.h)."""
    lib.require_hip(s, x)
    s, x = s.contiguous(), x.contiguous()
    n, D = x.shape
    F = update.node_dim
    C = update.node_irreps.num_irreps
    dev = s.device
    f32 = dict(dtype=torch.float32, device=dev)
    packed = packed_fwd(update, message)
    tail = message is not None
    assert want_x or not tail
    rows = int(lib.load().xeq_node_block_rows(n))   # internal tensors: whole workgroups, wave-native layout (include/xeq.h)
    o = {
        "p": torch.empty((rows, C), **f32), "uv": torch.empty(2 * rows * D, **f32), "stats": torch.empty((n, 4), **f32),
        "pre": torch.empty((rows, F), **f32), "a": torch.empty((rows, C + 2 * F), **f32), "ip": torch.empty((rows, F), **f32),
        "s_out": torch.empty((n, F), **f32), "x_out": torch.empty((n, D), **f32) if want_x else None,
    }
    if tail:
        o.update(stats2=torch.empty((n, 4), **f32), xhat2=torch.empty(n * D, **f32), pre2=torch.empty((rows, F), **f32),
                 h2=torch.empty((n, F + 2 * C), **f32))
    m = update.update_mlp
    nm = message
    call("xeq_node_block_fwd", n, ptr(s), ptr(x), ptr(update.norm.weight), ptr(update.norm.bias), ptr(update.o3norm.affine_weight),
         ptr(update.o3norm.affine_bias), ptr(_uv_bias(update)), ptr(m[0].bias), ptr(m[2].bias), float(update.invariant.eps), ptr(packed),
         ptr(o["p"]), ptr(o["uv"]), ptr(o["stats"]), ptr(o["pre"]), ptr(o["a"]), ptr(o["ip"]), ptr(o["s_out"]), ptr(o["x_out"]),
         ptr(nm.norm.weight if tail else None), ptr(nm.norm.bias if tail else None), ptr(nm.o3norm.affine_weight if tail else None),
         ptr(nm.o3norm.affine_bias if tail else None), ptr(nm.scalar_mlp[0].bias if tail else None),
         ptr(nm.scalar_mlp[2].bias if tail else None), ptr(o.get("stats2")), ptr(o.get("xhat2")), ptr(o.get("pre2")), ptr(o.get("h2")),
         stream())
    return o


def packed_bwd(update, message=None, with_gx: bool = True) -> torch.Tensor:
    """The reverse weight program of (update block, next message block or None), cached on the update module."""
    from .fused import _packed_uv

    m = update.update_mlp
    ws = [m[0].weight, update.update_U.weight, update.update_V.weight, update.dot_lin.weight, m[2].weight]
    if message is not None:
        ws += [message.scalar_mlp[0].weight, message.scalar_mlp[2].weight]
    with_gx = bool(with_gx or message is not None)
    key = (_versions(*ws), pack_epoch(), id(message), with_gx)
    cache = getattr(update, "_xeq_nb_bwd", None)
    if cache is None:
        cache = update._xeq_nb_bwd = {}
    hit = cache.get(with_gx)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        uv, _ = _packed_uv(update)
        tiles = int(lib.load().xeq_node_block_bwd_tiles(int(message is not None), int(with_gx)))
        out = torch.empty(tiles * TILE_BYTES, dtype=torch.uint8, device=m[0].weight.device)
        w1n = message.scalar_mlp[0].weight.detach().contiguous() if message is not None else None
        w2n = message.scalar_mlp[2].weight.detach().contiguous() if message is not None else None
        keep = [m[0].weight.detach().contiguous(), update.dot_lin.weight.detach().contiguous(), m[2].weight.detach().contiguous()]
        call("xeq_node_block_pack_bwd", ptr(keep[0]), ptr(uv[0]), ptr(uv[1]), ptr(uv[2]), ptr(keep[1]), ptr(keep[2]), ptr(w1n), ptr(w2n),
             int(with_gx), ptr(out), stream())
    cache[with_gx] = (key, out)
    return out


def node_block_bwd(saved: dict, s: torch.Tensor, x: torch.Tensor, update, message, g_s_in: torch.Tensor, g_x_in: Optional[torch.Tensor],
                   g_h: Optional[torch.Tensor] = None, g_xhat: Optional[torch.Tensor] = None):
    """Reverse of ``node_block_fwd`` (input gradients).  ``saved``: what the forward launch returned.  With the next block's front
    half: g_h, g_xhat (BT) are the gradients of h2 / xhat2 and g_s_in / g_x_in those reaching s_out / x_out directly."""
    n, D = x.shape
    F = update.node_dim
    C = update.node_irreps.num_irreps
    f32 = dict(dtype=torch.float32, device=s.device)
    tail = message is not None
    assert tail == (g_h is not None)
    packed = packed_bwd(update, message, with_gx=g_x_in is not None)
    g_s, g_x = torch.empty((n, F), **f32), torch.empty((n, D), **f32)
    rows = int(lib.load().xeq_node_block_rows(n))
    gxo = torch.empty((rows, D), **f32) if (tail or g_x_in is not None) else None
    gp, gv, gw = torch.empty((rows, C), **f32), torch.empty((rows, C), **f32), torch.empty((rows, D), **f32)
    # contiguous copies stay bound to names until the launch is enqueued: a temporary freed after its pointer was taken could be
    # handed by the caching allocator to the next copy
    g_h, g_xhat, g_s_in, g_x_in = (None if t is None else t.contiguous() for t in (g_h, g_xhat, g_s_in, g_x_in))
    call("xeq_node_block_bwd", n, ptr(g_h), ptr(g_xhat), ptr(g_s_in), ptr(g_x_in), ptr(saved["s_out"] if tail else None),
         ptr(saved["x_out"] if tail else None), ptr(saved.get("stats2")), ptr(saved.get("pre2")),
         ptr(message.norm.weight if tail else None), ptr(message.o3norm.affine_weight if tail else None), ptr(saved["uv"]), ptr(saved["a"]),
         ptr(saved["ip"]), ptr(saved["pre"]), ptr(s), ptr(x), ptr(saved["stats"]), ptr(update.norm.weight), ptr(update.o3norm.affine_weight),
         float(update.invariant.eps), ptr(packed), ptr(gxo), ptr(gp), ptr(gv), ptr(gw), ptr(g_s), ptr(g_x), stream())
    return g_s, g_x


def native_to_rows(buf: torch.Tensor, n: int, width: int) -> torch.Tensor:
    k = width // 32
    wb = buf.numel() // (k * 512)
    t = buf.reshape(wb, k, 2, 4, 16, 4).permute(0, 4, 1, 2, 3, 5).reshape(wb * 16, width)
    return t[:n]
