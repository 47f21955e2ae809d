"""Operator-level drop-ins for the ``e3nn.o3`` pieces on the XPaiNN path.

Same constructor / call signatures as the e3nn classes the reference uses
(SURVEY 8b "operator-level"), backed by the HIP kernels of libxeq_hip.so:

=============================  =============================================
reference call site            replacement
=============================  =============================================
nn/xpainn.py:49-51, 71-74      :class:`SphericalHarmonics`
nn/xpainn.py:119-121, 150-153  :class:`ElementwiseTensorProduct`
nn/o3layer.py:27-29, 93-95     :class:`TensorProduct` (the 'uuu' l x l -> 0e form)
nn/xpainn.py:186-187           :class:`Linear`
=============================  =============================================

Only what the path needs is implemented: irreps with l <= 2, blocks in
ascending l, at most one block per l.  Anything else raises NotImplementedError.
"""
from __future__ import annotations

import math
import re
from typing import Iterable, List, Tuple, Union

import torch
import torch.nn as nn

from . import ops


class Irrep(tuple):
    """(l, p)"""

    def __new__(cls, l: int, p: int):
        return super().__new__(cls, (int(l), int(p)))

    def __getnewargs__(self):   # copy.deepcopy / pickle of modules that hold irreps (tuple subclasses rebuild through __new__)
        return (self[0], self[1])

    @property
    def l(self) -> int:
        return self[0]

    @property
    def p(self) -> int:
        return self[1]

    @property
    def dim(self) -> int:
        return 2 * self[0] + 1

    def __repr__(self) -> str:
        return f"{self.l}{'e' if self.p == 1 else 'o'}"


class Irreps:
    """Minimal stand-in for ``e3nn.o3.Irreps``: parsing, ``dim``, ``num_irreps``,
    iteration over ``(mul, Irrep)``, ``simplify`` and ``==``."""

    def __init__(self, irreps: Union[str, "Irreps", Iterable]):
        if isinstance(irreps, Irreps):
            self._items: List[Tuple[int, Irrep]] = list(irreps._items)
        elif isinstance(irreps, str):
            self._items = []
            for term in irreps.split("+"):
                m = re.fullmatch(r"\s*(?:(\d+)\s*x\s*)?(\d+)\s*([eo])\s*", term)
                if m is None:
                    raise ValueError(f"Unable to convert string {irreps!r} into an Irreps")
                self._items.append((int(m.group(1) or 1), Irrep(int(m.group(2)), 1 if m.group(3) == "e" else -1)))
        else:
            self._items = []
            for item in irreps:
                mul, ir = item
                if not isinstance(ir, Irrep):
                    ir = Irrep(*ir) if not isinstance(ir, str) else Irreps(ir)._items[0][1]
                self._items.append((int(mul), ir))

    def __iter__(self):
        return iter(self._items)

    def __len__(self) -> int:
        return len(self._items)

    def __getitem__(self, i):
        return self._items[i]

    def __eq__(self, other) -> bool:
        return self._items == Irreps(other)._items

    def __repr__(self) -> str:
        return "+".join(f"{mul}x{ir}" for mul, ir in self._items)

    @property
    def dim(self) -> int:
        return sum(mul * ir.dim for mul, ir in self._items)

    @property
    def num_irreps(self) -> int:
        return sum(mul for mul, _ in self._items)

    @property
    def lmax(self) -> int:
        return max(ir.l for _, ir in self._items)

    def simplify(self) -> "Irreps":
        out: List[Tuple[int, Irrep]] = []
        for mul, ir in self._items:
            if mul == 0:
                continue
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            else:
                out.append((mul, ir))
        return Irreps(out)

    def mul3(self) -> Tuple[int, int, int]:
        """Kernel descriptor: channels of l = 0, 1, 2 (blocks ascending in l)."""
        mul = [0, 0, 0]
        last = -1
        for m, ir in self.simplify():
            if ir.l > 2:
                raise NotImplementedError(f"irreps {self}: l = {ir.l} > 2 is not supported by the HIP kernels")
            if ir.l <= last:
                raise NotImplementedError(f"irreps {self}: blocks must be in ascending l, one per l")
            mul[ir.l] = m
            last = ir.l
        return tuple(mul)

    def blocks(self):
        """(mul, l, flat_offset, channel_offset) per block."""
        off = ch = 0
        for mul, ir in self._items:
            yield mul, ir.l, off, ch
            off += mul * ir.dim
            ch += mul


class SphericalHarmonics(nn.Module):
    """``e3nn.o3.SphericalHarmonics(irreps_out, normalize, normalization)`` for
    ``normalization='component'``; every Y_l is repeated ``mul`` times
    (nn/xpainn.py:49-51).  Input in e3nn axis order."""

    def __init__(self, irreps_out, normalize: bool, normalization: str = "integral") -> None:
        super().__init__()
        if normalization != "component":
            raise NotImplementedError("only normalization='component' is on the XPaiNN path")
        self.irreps_out = Irreps(irreps_out)
        self.normalize = bool(normalize)
        self._mul = self.irreps_out.mul3()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        out = ops.SphHarm.apply(x.reshape(-1, 3), self._mul, self.normalize)
        return out.reshape(*lead, out.shape[-1])


class ElementwiseTensorProduct(nn.Module):
    """``e3nn.o3.ElementwiseTensorProduct(irreps_in1, "Cx0e")`` (nn/xpainn.py:119-121):
    out[u,m] = x[u,m] * g[u]."""

    def __init__(self, irreps_in1, irreps_in2) -> None:
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        ok = all(ir.l == 0 and ir.p == 1 for _, ir in self.irreps_in2)
        if not ok or self.irreps_in2.num_irreps != self.irreps_in1.num_irreps:
            raise NotImplementedError("only the (irreps, 'Cx0e') gating form is on the XPaiNN path")
        self.irreps_out = self.irreps_in1
        self._mul = self.irreps_in1.mul3()

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return ops.ElementwiseTP.apply(x, y, self._mul)


class TensorProduct(nn.Module):
    """The one ``e3nn.o3.TensorProduct`` form on the path: instructions
    ``(i, i, i, 'uuu', False, ir.dim)`` into ``mul x 0e`` with component
    normalisation (nn/o3layer.py:23-29, 89-95) = per-channel dot product."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, irrep_normalization: str = "component") -> None:
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)
        if self.irreps_in1 != self.irreps_in2 or irrep_normalization != "component":
            raise NotImplementedError("only the self 'uuu' -> 0e product is on the XPaiNN path")
        for k, ins in enumerate(instructions):
            i1, i2, io, mode, has_w, pw = ins
            if not (i1 == i2 == io == k and mode == "uuu" and not has_w and pw == self.irreps_in1[k][1].dim):
                raise NotImplementedError(f"unsupported TensorProduct instruction {ins}")
        self._mul = self.irreps_in1.mul3()

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return ops.ChannelDot.apply(x, y, self._mul)


class Linear(nn.Module):
    """``e3nn.o3.Linear(irreps_in, irreps_out, biases=True)`` for equal in/out irreps
    (nn/xpainn.py:186-187): out[w,m] = mul^-1/2 sum_u W_l[u,w] x[u,m]; flat weight =
    concat of row-major [mul, mul] blocks, ~N(0,1); bias (zeros) on 0e only."""

    def __init__(self, irreps_in, irreps_out, biases: bool = False) -> None:
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_out)
        if self.irreps_in != self.irreps_out:
            raise NotImplementedError("only irreps_in == irreps_out is on the XPaiNN path")
        self._mul = self.irreps_in.mul3()
        n_w = sum(mul * mul for mul, _ in self.irreps_in)
        self.weight = nn.Parameter(torch.randn(n_w))
        n_b = sum(mul for mul, ir in self.irreps_in if ir.l == 0 and ir.p == 1) if biases else 0
        self.bias = nn.Parameter(torch.zeros(n_b))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops.lib.require_hip(x)
        n = x.shape[0]
        parts = []
        woff = 0
        for mul, l, off, _ in self.irreps_in.blocks():
            d = 2 * l + 1
            W = self.weight[woff : woff + mul * mul].view(mul, mul)
            woff += mul * mul
            xb = x[:, off : off + mul * d]
            if d == 1:
                ob = torch.addmm(self.bias, xb, W, alpha=1.0 / math.sqrt(mul)) if self.bias.numel() > 0 else torch.mm(xb, W) * (1.0 / math.sqrt(mul))
            else:
                # one plain [N*d, mul] x [mul, mul] GEMM on a contiguous m-major copy (a strided
                # batched GEMM over N tiny [d, mul] panels is ~10x slower on rocBLAS)
                xt = xb.reshape(n, mul, d).transpose(1, 2).reshape(n * d, mul)
                ob = torch.mm(xt, W).reshape(n, d, mul).transpose(1, 2).reshape(n, mul * d) * (1.0 / math.sqrt(mul))
            parts.append(ob)
        return torch.cat(parts, dim=-1)
