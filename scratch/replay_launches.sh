#!/bin/bash
# usage (GPU box): scratch/replay_launches.sh  -- kernels of ONE replayed whole-step graph: call counts of two profiled bench runs
# (K = 10 and K = 110 timed steps, same warm-up and host-launched legs) differenced
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for k in 10 110; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rl_$k -- python3 $R/bench.py --steps $k --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_rl_$k.log 2>&1
  f=$(find $R/gpurun_out/prof_rl_$k -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/rl_$k.csv; rm -rf $R/gpurun_out/prof_rl_$k
done
cd $R
python3 - <<'PY'
import csv
a = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("gpurun_out/rl_10.csv"))}
b = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open("gpurun_out/rl_110.csv"))}
rows = []
for n, (c, t) in b.items():
    c0, t0 = a.get(n, (0, 0.0))
    if c - c0 > 0:
        rows.append(((c - c0) / 100, (t - t0) / 100 / 1e3, n))
rows.sort(key=lambda r: -r[1])
print(f"launches per replayed step {sum(r[0] for r in rows):.1f}; kernel time {sum(r[1] for r in rows):.1f} us")
for c, t, n in rows:
    print(f"{c:6.2f} {t:8.1f} us  {n[:120]}")
PY
