"""Per-kernel totals of a rocprofv3 kernel trace directory divided by the number of timed steps: argv[1] dir, argv[2] steps (the first
occurrences -- eager warm-up and capture -- are dropped by taking the LAST steps * count-per-step instances of every name)."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return re.sub(r"^void ", "", n).split("(")[0][:70]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0; nk = 0
first = {}
for i, r in enumerate(rows[-len(rows) // 2:]):
    first.setdefault(short(r["Kernel_Name"]), i)
for k, v in sorted(by.items(), key=lambda kv: first.get(kv[0], 1 << 30)):
    per = len(v) // (steps + 5)          # instances per step (the trace also holds 5 warm-up steps, eager or replayed)
    if per == 0: continue
    last = v[-per * steps:]
    t = sum(last) / steps
    tot += t; nk += per
    print(f"  {per:3d} x {sum(last) / len(last):7.2f} us = {t:7.1f} us  {k}")
print(f"  {nk} kernels per step, kernel time {tot:.1f} us")
