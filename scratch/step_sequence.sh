#!/bin/bash
# usage (GPU box): scratch/step_sequence.sh <tag>  -- the kernels of ONE replayed whole-step graph of bench.py, in launch order
# (kernel trace; steps are cut at k_load_padded_batch, the most frequent step length is taken as the replayed one)
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/seq_$tag -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/seq_$tag.log 2>&1
cd $R
python3 - <<PY > gpurun_out/${tag}_step_sequence.txt
import csv, glob, re, collections
f = glob.glob("gpurun_out/seq_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n)
    if "at::native" in n or "rocprim" in n or "at::cuda" in n or "elementwise" in n:
        m = re.search(r"(rocprim::detail::\w+|at::native::\w+<[^,>]*|\w+_kernel\w*)", n)
        return "ATen/rocprim: " + (m.group(1) if m else n[:60])
    return n.split("(")[0][:70]
cuts = [i for i, r in enumerate(rows) if "k_load_padded_batch" in r["Kernel_Name"]] + [len(rows)]
steps = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
lens = collections.Counter(len(s) for s in steps)
L = lens.most_common(1)[0][0]
s = [x for x in steps if len(x) == L][-1]
t0 = int(s[0]["Start_Timestamp"])
print(f"one replayed whole-step graph: {L} kernels ({lens[L]} of {len(steps)} steps in the trace have this length), {(int(s[-1]['End_Timestamp']) - t0) / 1e3:.0f} us from first start to last end")
for i, r in enumerate(s):
    print(f"{i + 1:3d}  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
PY
rm -rf gpurun_out/seq_$tag
cat gpurun_out/${tag}_step_sequence.txt
