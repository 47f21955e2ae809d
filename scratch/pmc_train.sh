#!/bin/bash
# usage (GPU box): scratch/pmc_train.sh  -- HBM traffic (FETCH_SIZE, WRITE_SIZE in separate --pmc passes, kernel trace only) of the own kernels of the
# energy+force training step (scratch/bench_train.py 1024 forces) -> gpurun_out/r04_train_traffic_pmc.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_train_$c -- python3 $R/scratch/bench_train.py 256 forces > $R/gpurun_out/pmc_train_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "xeq::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
fe, wr = collect("/tmp/pmc_train_FETCH_SIZE", "FETCH_SIZE"), collect("/tmp/pmc_train_WRITE_SIZE", "WRITE_SIZE")
with open("gpurun_out/r04_train_traffic_pmc.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean_raw", "FETCH_x2_MB", "WRITE_SIZE_KiB_mean", "HBM_MB_per_launch(2*F+W)"])
    for k in sorted(set(fe) | set(wr), key=lambda k: -sum(fe.get(k, [0]))):
        f = sum(fe.get(k, [0])) / max(1, len(fe.get(k, []))); x = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [])))
        w.writerow([k[:110], len(fe.get(k, [])), f"{f:.1f}", f"{2 * f * 1024 / 1e6:.2f}", f"{x:.1f}", f"{(2 * f + x) * 1024 / 1e6:.2f}"])
PY
grep "train step" gpurun_out/pmc_train_FETCH_SIZE.log | head -2
cat gpurun_out/r04_train_traffic_pmc.csv
