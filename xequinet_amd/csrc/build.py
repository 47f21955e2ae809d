"""Build libxeq_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m xequinet_amd.csrc.build [--force]

The shared library is written IN-TREE (xequinet_amd/libxeq_hip.so) so that it
travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libxeq_hip.so")
SOURCES = ["xeq_graph.hip", "xeq_ops.hip", "xeq_message.hip", "xeq_message_sb.hip", "xeq_message_wq.hip", "xeq_message_wq_bwd.hip", "xeq_node.hip", "xeq_mlp.hip", "xeq_linear.hip", "xeq_update.hip", "xeq_nodeblock.hip", "xeq_tp.hip", "xeq_train.hip", "xeq_train_node.hip"]
HEADERS = ["xeq_common.h", os.path.join("..", "..", "include", "xeq.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]
# per-source extras.  The matrix-core message kernels: LLVM's max-ILP machine scheduler instead of the default (measured in round 1 on
# their predecessor: reverse launch 573 -> 544 us, same VGPR budgets);
EXTRA_FLAGS = {# node block: no packed-fp32 instructions.  With two waves of this kernel on a SIMD, v_pk_fma_f32 / v_pk_add_f32 results computed
               # from matrix-core outputs came out wrong in one 16-lane row, sporadically (profiles/r04_nodeblock.txt item 9c: bisected to exactly this)
               "xeq_nodeblock.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               # wq: explicit fma chains only (its window / global instantiations must round alike)
               "xeq_message_wq.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-ffp-contract=off"],
               # its reverse half (same text, XEQ_WQ_PART_BWD): the default scheduler orders the unfenced reverse tile better
               "xeq_message_wq_bwd.hip": ["-ffp-contract=off"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        extra_deps = [os.path.join(HERE, "xeq_message_wq.hip")] if src == "xeq_message_wq_bwd.hip" else []
        if force or _stale(o, [s] + hdrs + extra_deps):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
