#!/bin/bash
# usage (GPU box): scratch/prof_tp.sh  -- rocprofv3 kernel stats of scratch/bench_tp.py -> gpurun_out/r04_tp_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tp -o p -- python3 $R/scratch/bench_tp.py > $R/gpurun_out/prof_tp.log 2>&1
cd $R
f=$(find gpurun_out/prof_tp -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "k_tp_" in r["Name"]]
lib = [r for r in rows if r["Name"].startswith("Cijk")]
with open("gpurun_out/r04_tp_kernel_stats.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_us", "average_us"])
    for r in keep:
        w.writerow([r["Name"].split("(")[0], r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.1f}", f"{float(r['AverageNs']) / 1e3:.1f}"])
    if lib:
        w.writerow([f"library GEMMs behind the einsum weight gradients of the per-path form ({len(lib)} kernels)", sum(int(r["Calls"]) for r in lib),
                    f"{sum(float(r['TotalDurationNs']) for r in lib) / 1e3:.1f}", ""])
PY
rm -rf gpurun_out/prof_tp
cat gpurun_out/r04_tp_kernel_stats.csv | cut -c1-200
