#!/bin/bash
# usage (GPU box): scratch/prof_md.sh <system> -- per-step kernel table of the LAMMPS-style replay
sysn=$1; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profmd_$sysn -o p -- python3 $R/scratch/prof_md.py $sysn 200 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, re
f = glob.glob("gpurun_out/profmd_$sysn/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 200.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$sysn: kernel time per step %.1f us (incl. warm-up / capture launches)" % (tot / steps / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    n = re.sub(r"^void ", "", r["Name"]).split("(")[0][:70]
    print("%8.1f us/step %6.2f launches/step %7.1f us each  %s" % (float(r["TotalDurationNs"]) / steps / 1e3, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, n))
PY
rm -rf gpurun_out/profmd_$sysn
