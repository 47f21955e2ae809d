#!/bin/bash
# usage (GPU box): scratch/idle_train.sh [mode]  -- GPU idle share inside the training steps of scratch/bench_train.py (kernel trace): where the host is the bound
R=$GRAFT_REPO_ROOT
MODE=${1:-forces}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/idle_train
rocprofv3 --kernel-trace --output-format csv -d /tmp/idle_train -o p -- python3 $R/scratch/bench_train.py 1024 $MODE > $R/gpurun_out/idle_train.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/idle_train/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the timed region: the last 15 steps (10 timed + 5 with the launch timer); cut at the optimizer's multi_tensor_apply kernels (end of a step)
ends = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
# group consecutive optimizer kernels
marks = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] != ends[i] + 1]
steps = [(a, b) for a, b in ((marks[i] + 1, marks[i + 1]) for i in range(len(marks) - 1)) if rows[b][1] - rows[a][0] < 80e6][-10:]
tot_span = tot_busy = 0
gaps = []
for a, b in steps:
    span = rows[b][1] - rows[a][0]
    busy = sum(r[1] - r[0] for r in rows[a : b + 1])
    tot_span += span; tot_busy += busy
    for i in range(a, b):
        g = rows[i + 1][0] - rows[i][1]
        if g > 10000: gaps.append((g, rows[i][2][:60], rows[i + 1][2][:60]))
print(f"{len(steps)} steps: span {tot_span / len(steps) / 1e6:.2f} ms, kernels busy {tot_busy / len(steps) / 1e6:.2f} ms, idle {100 * (1 - tot_busy / tot_span):.1f} %")
gaps.sort(reverse=True)
print("largest gaps (us, after kernel, before kernel):")
for g, a, b in gaps[:12]: print(f"  {g / 1e3:8.1f}  {a}  ->  {b}")
print("gaps > 10 us per step:", len(gaps) / len(steps), "their sum per step (ms):", sum(g for g, _, _ in gaps) / len(steps) / 1e6)
PY
