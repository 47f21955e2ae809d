"""Per-evaluation table from a rocprofv3 kernel_stats.csv of bench.py: python scratch/stats_table.py <csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sum(int(r["Calls"]) for r in rows if "k_message_fwd" in r["Name"]) / 3   # three message blocks per evaluation
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"evaluations {ev:.0f}; kernel time per evaluation {tot / ev / 1e3:.1f} us")
g = {}
for r in rows:
    n = r["Name"]
    k = "library GEMMs (Cijk)" if n.startswith("Cijk") else ("ATen / rocprim / copies" if ("at::native" in n or "rocprim" in n or "rocclr" in n or "compute_cuda" in n or "elementwise" in n or "at::cuda" in n) else re.sub(r"^void ", "", n).split("(")[0][:60])
    a = g.setdefault(k, [0, 0]); a[0] += float(r["TotalDurationNs"]) / ev / 1e3; a[1] += int(r["Calls"]) / ev
for k, v in sorted(g.items(), key=lambda x: -x[1][0]):
    print(f"{v[0]:8.1f} us {v[1]:6.1f} launches  {k}")
print(f"launches per evaluation {sum(v[1] for v in g.values()):.1f}")
