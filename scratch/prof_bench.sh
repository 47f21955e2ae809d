#!/bin/bash
# usage (GPU box): scratch/prof_bench.sh <tag>  -- rocprofv3 kernel stats of bench.py, per-eval table to gpurun_out/prof_<tag>.txt
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
XEQ_NO_GEMM_TUNING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --steps 40 --warmup 5 > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
cd $R
python3 - <<PY > gpurun_out/prof_$tag.txt
import csv, glob, json, re
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if not r["Name"].startswith("Cijk") or True]
calls = {r["Name"]: int(r["Calls"]) for r in rows}
ev = max(c for n, c in calls.items() if "k_message_fwd" in n) / 3
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("evaluations", ev, "kernel time per evaluation us", tot / ev / 1e3)
g = {}
for r in rows:
    n = r["Name"]
    k = "gemm (Cijk)" if n.startswith("Cijk") else ("aten / rocprim / copies" if ("at::native" in n or "rocprim" in n or "rocclr" in n or "compute_cuda" in n or "elementwise" in n) else re.sub(r"^void ", "", n).split("(")[0][:60])
    a = g.setdefault(k, [0, 0]); a[0] += float(r["TotalDurationNs"]) / ev / 1e3; a[1] += int(r["Calls"]) / ev
for k, v in sorted(g.items(), key=lambda x: -x[1][0]):
    print(f"{v[0]:8.1f} us {v[1]:6.1f} launches  {k}")
print("launches per evaluation", sum(v[1] for v in g.values()))
PY
rm -rf gpurun_out/prof_$tag
cat gpurun_out/prof_$tag.txt
