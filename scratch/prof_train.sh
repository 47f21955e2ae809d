#!/bin/bash
# usage (GPU box): scratch/prof_train.sh [mode]  -- rocprofv3 kernel stats of scratch/bench_train.py -> gpurun_out/train_kernel_stats.csv
R=$GRAFT_REPO_ROOT
MODE=${1:-forces}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python3 $R/scratch/bench_train.py ${NMOL:-1024} $MODE > $R/gpurun_out/prof_train.log 2>&1
cd $R
f=$(find /tmp/prof_train -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 30.0   # 15 warm-up + 10 timed + 5 with the launch timer (scratch/bench_train.py)
with open("gpurun_out/train_kernel_stats.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls_per_step", "us_per_step", "average_us", "percent"])
    for r in rows[:60]:
        w.writerow([r["Name"][:110], f"{int(r['Calls']) / steps:.1f}", f"{float(r['TotalDurationNs']) / 1e3 / steps:.1f}", f"{float(r['AverageNs']) / 1e3:.1f}", f"{float(r['TotalDurationNs']) / tot * 100:.1f}"])
    w.writerow(["ALL", f"{sum(int(r['Calls']) for r in rows) / steps:.1f}", f"{tot / 1e3 / steps:.1f}", "", "100"])
PY
tail -2 gpurun_out/prof_train.log
