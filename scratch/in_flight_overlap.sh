#!/bin/bash
# usage (GPU box): scratch/in_flight_overlap.sh  -- kernel trace of bench.py (two steps in flight): how much of the replayed region has 0 / 1 / 2+ kernels running
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ovl -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ovl.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ovl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the replayed region: the longest run of steps cut at k_load_padded_batch with <= 35 kernels between cuts on one queue
loads = [i for i, r in enumerate(rows) if "k_load_padded_batch" in r[2]]
# take the 30 timed steps: the last 30 + 30 (one-at-a-time) loads are behind; find the window where loads come from two queues alternately
qs = [rows[i][3] for i in loads]
best = None
for a in range(len(loads) - 30):
    w = qs[a:a + 30]
    if len(set(w)) == 2 and all(w[k] != w[k + 1] for k in range(29)):
        best = a
if best is None:
    print("no alternating window found; queues seen:", sorted(set(qs))); raise SystemExit
t0, t1 = rows[loads[best + 2]][0], rows[loads[best + 28]][0]
ev = []
for s, e, n, q in rows:
    if e <= t0 or s >= t1: continue
    ev.append((max(s, t0), 1)); ev.append((min(e, t1), -1))
ev.sort()
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
hist[depth] = hist.get(depth, 0) + (t1 - last)
tot = float(t1 - t0)
print(f"window of 26 steps: {tot / 26 / 1e3:.1f} us per step; time with k kernels running: " + ", ".join(f"{k}: {v / tot * 100:.1f} %" for k, v in sorted(hist.items())))
PY
rm -rf gpurun_out/ovl
