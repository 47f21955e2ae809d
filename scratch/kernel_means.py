"""Mean duration per kernel name over the steady-state steps of a rocprofv3 kernel trace directory (argv[1])."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if "k_load_padded_batch" in r["Kernel_Name"]] + [len(rows)]
steps = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
L = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == L][5:-1]
def short(n):
    return re.sub(r"^void ", "", n).split("(")[0][:60]
tot = collections.OrderedDict()
for s in steps:
    for r in s:
        k = short(r["Kernel_Name"])
        d = tot.setdefault(k, [0, 0.0])
        d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = sum((int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3 for s in steps) / len(steps)
print(f"{len(steps)} steady steps of {L} kernels, {span:.1f} us first start to last end")
for k, (c, t) in tot.items():
    print(f"  {c // len(steps):3d} x {t / c:7.2f} us = {t / len(steps):7.1f} us  {k}")
print(f"  TOTAL kernel time {sum(t for c, t in tot.values()) / len(steps):.1f} us")
