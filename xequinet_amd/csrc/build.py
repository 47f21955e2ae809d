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
HEADERS = ["xeq_common.h", "xeq_linear_s.h", os.path.join("..", "..", "include", "xeq.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]
# per-source extras.  The matrix-core message kernels: LLVM's max-ILP machine scheduler instead of the default (measured in round 1 on
# their predecessor: reverse launch 573 -> 544 us, same VGPR budgets).
# No packed-fp32 instructions in any object that holds matrix-core kernels.  With two waves of the node-block kernel on a SIMD,
# v_pk_fma_f32 / v_pk_add_f32 results computed from matrix-core outputs came out wrong in one 16-lane row, sporadically
# (profiles/r04_nodeblock.txt item 9c: bisected to exactly this; reproducer scratch/ubench/pk_after_mfma.hip).  The cause is not
# pinned to the compiler or the silicon, the scalar forms cost nothing measurable (profiles/r04_small_experiments.txt item 1,
# profiles/r05_no_packed.txt), so every object with v_mfma in it is built without them and `check_no_packed` fails the build if the
# flag is ever dropped by a toolchain update.
NO_PACKED = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
MFMA_SOURCES = ["xeq_nodeblock.hip", "xeq_message_wq.hip", "xeq_message_wq_bwd.hip", "xeq_update.hip", "xeq_mlp.hip", "xeq_linear.hip", "xeq_train.hip"]
EXTRA_FLAGS = {"xeq_nodeblock.hip": [],
               # wq: explicit fma chains only (its window / global instantiations must round alike)
               "xeq_message_wq.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-ffp-contract=off"],
               # its reverse half (same text, XEQ_WQ_PART_BWD): the default scheduler orders the unfenced reverse tile better
               "xeq_message_wq_bwd.hip": ["-ffp-contract=off"]}
for _src in MFMA_SOURCES:
    EXTRA_FLAGS[_src] = EXTRA_FLAGS.get(_src, []) + NO_PACKED
LLVM_BIN = os.environ.get("XEQ_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def device_isa(obj):
    """Disassembly (text) of the gfx950 code object inside a host object file built by hipcc."""
    import shutil
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        objdump = os.path.join(LLVM_BIN, "llvm-objdump")
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True)     # writes <obj>.0.<target> next to its input
        parts = [f for f in os.listdir(tmp) if "gfx950" in f]
        if len(parts) != 1:
            raise RuntimeError(f"{obj}: expected one gfx950 code object, found {parts}")
        return subprocess.run([objdump, "-d", os.path.join(tmp, parts[0])], check=True, capture_output=True, text=True).stdout


def packed_fp32_counts(src):
    """(v_pk_{fma,add,mul}_f32 instructions, v_mfma instructions) in the built object of one source file."""
    import re

    isa = device_isa(os.path.join(HERE, "build", src.replace(".hip", ".o")))
    return len(re.findall(r"\bv_pk_(?:fma|add|mul)_f32\b", isa)), len(re.findall(r"\bv_mfma_", isa))


def check_no_packed(sources=None):
    """Fail if a matrix-core object contains a packed-fp32 instruction, or if a source outside MFMA_SOURCES has grown matrix-core
    kernels (it would then need the flag too)."""
    for src in sources or SOURCES:
        pk, mfma = packed_fp32_counts(src)
        if src in MFMA_SOURCES and pk:
            raise RuntimeError(f"{src}: {pk} packed-fp32 instructions in a matrix-core object (is -packed-fp32-ops still honoured?)")
        if src not in MFMA_SOURCES and mfma:
            raise RuntimeError(f"{src}: {mfma} v_mfma instructions but the file is not in build.MFMA_SOURCES")


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
    if jobs:
        check_no_packed([os.path.basename(j[-3]) for j in jobs])
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
