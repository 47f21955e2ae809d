#!/bin/bash
# usage (GPU box): scratch/seq_water.sh <tag> [workload]  -- the kernels of one replayed whole-step graph of bench.py --workload water_512, in launch order
tag=${1:-w}; wl=${2:-water_512}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/seq_$tag -- python3 $R/bench.py --workload $wl --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/seq_$tag.log 2>&1
cd $R
python3 - <<PY > gpurun_out/${tag}_step_sequence.txt
import csv, glob, re, collections
f = glob.glob("gpurun_out/seq_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n)
    if "at::native" in n or "rocprim" in n or "at::cuda" in n or "elementwise" in n or "hipcub" in n:
        m = re.search(r"(rocprim::detail::\w+|hipcub::\w+|at::native::\w+<[^,>]*|\w+_kernel\w*)", n)
        return "ATen/rocprim: " + (m.group(1) if m else n[:60])
    return n.split("(")[0][:70]
# the last 12 steps are replays of one graph: find the period by the most frequent kernel that occurs once per step
names = [r["Kernel_Name"] for r in rows]
key = next(n for n in reversed(names) if "k_edge_vectors_bwd" in n)
idx = [i for i, n in enumerate(names) if n == key]
a, b = idx[-3], idx[-2]
s = rows[a + 1 : b + 1]
t0 = int(s[0]["Start_Timestamp"])
print(f"one replayed step ($wl): {len(s)} kernels, {(int(s[-1]['End_Timestamp']) - t0) / 1e3:.0f} us from first start to last end")
for i, r in enumerate(s):
    print(f"{i + 1:3d}  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
PY
rm -rf gpurun_out/seq_$tag
cat gpurun_out/${tag}_step_sequence.txt
