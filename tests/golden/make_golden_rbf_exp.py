#!/usr/bin/env python
"""Fixtures of the reference's exponential radial bases (xequinet/nn/rbf.py:161-207: ExponentialBernstein, ExponentialNorm), generated
by importing that one module (it needs math, numpy and torch only) from /root/reference -- run in the build container, never on the
GPU box.  Writes rbf_exp_f32.npz / rbf_exp_f64.npz next to this script: the basis values on a distance grid and their derivatives with
respect to the distance (the reference's autograd), for
  * resolve_rbf("expbern", 20, 5.0)   -- which hands the CUTOFF over as alpha (rbf.py:15 against :162): alpha = 5.0,
  * ExponentialBernstein(20)           -- the class default, alpha = 0.5,
  * resolve_rbf("expnorm", 20, 5.0).
Data only: inputs and the reference's outputs."""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location("ref_rbf", f"{REF}/xequinet/nn/rbf.py")
    rbf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rbf)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        torch.set_default_dtype(dt)
        d = torch.cat([torch.linspace(0.3, 6.0, 58), torch.tensor([0.5, 2.0, 4.999, 5.0, 5.0001])]).to(dt).view(-1, 1)
        out = {"dist": d.numpy()}
        for name, mod in (("expbern20_a5", rbf.resolve_rbf("expbern", 20, 5.0)), ("expbern20_a05", rbf.ExponentialBernstein(20)),
                          ("expnorm20_rc5", rbf.resolve_rbf("expnorm", 20, 5.0))):
            x = d.clone().requires_grad_()
            y = mod(x)
            out[name] = y.detach().numpy()
            # d rho_k / d dist per basis function: one reverse pass per column
            out[name + "_ddist"] = np.stack([torch.autograd.grad(y[:, k].sum(), x, retain_graph=True)[0].reshape(-1).numpy()
                                             for k in range(y.shape[1])], axis=1)
            for pname, p in mod.named_parameters():
                out[f"{name}_param_{pname}"] = p.detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"rbf_exp_{tag}.npz"), **out)
    torch.set_default_dtype(torch.float32)


if __name__ == "__main__":
    main()
