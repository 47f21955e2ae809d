"""Build libxeq_torch.so: the TorchScript-visible operators (csrc/xeq_torch.cpp) over libxeq_hip.so.

    python -m xequinet_amd.csrc.build_torch [--force]

Written IN-TREE next to libxeq_hip.so (rpath $ORIGIN) so that both travel with the repository snapshot.
"""
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SRC = os.path.join(HERE, "xeq_torch.cpp")
LIB = os.path.join(PKG, "libxeq_torch.so")


def build(force=False, verbose=True):
    from .build import _hipcc, _stale, build as build_hip

    hip_lib = build_hip(verbose=verbose)
    if not force and not _stale(LIB, [SRC, hip_lib, os.path.join(HERE, "..", "..", "include", "xeq.h"), os.path.abspath(__file__)]):
        return LIB
    from torch.utils import cpp_extension as ce

    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-unused-function"]
    for inc in ce.include_paths():
        cmd += ["-I", inc]
    cmd += ["-I", "/opt/rocm/include", SRC, "-o", LIB, "-L", tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
            "-L", PKG, "-l:libxeq_hip.so", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
