"""Summarise the rocprofv3 --pmc pass of profiles/collect_sq.sh: per kernel, the mean of every SQ counter per launch, the launch
duration from the kernel trace of the same run, and two derived figures:

  per-wave matrix share = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)   (MFMA-busy counts cycles, SQ_WAVE_CYCLES quad-cycles:
      MI355X_MICROARCH.md, counter table) -- the share of a wave's lifetime in which its SIMD's matrix pipe works for it;
  chip matrix busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz) -- against the whole chip's pipes for the
      launch (the clock under load is 2.1-2.4 GHz: read it as +-10 %).

The ratio the round-3 review asked for, SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, is printed too; SQ_BUSY_CYCLES is summed per shader
engine, not per SIMD, so that ratio exceeds 1 for a kernel that keeps several SIMDs' pipes busy."""
import csv, glob, os, sys, collections

tag, d = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_INSTS_MFMA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
rows = []
for k, cs in acc.items():
    if "xeq" not in k:
        continue
    m = {c: (sum(cs[c]) / len(cs[c]) if cs.get(c) else 0.0) for c in names}
    n = len(next(iter(cs.values())))
    us = sum(dur[k]) / max(1, len(dur[k])) / 1e3 if dur.get(k) else 0.0
    rows.append((k, n, us, m))
rows.sort(key=lambda r: -r[1] * r[2])
out = os.path.join("gpurun_out", f"{tag}_sq_counters.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches", "us_per_launch_under_pmc"] + names + ["per_wave_matrix_share", "chip_matrix_busy", "mfma_busy_over_sq_busy", "wait_any_share", "wait_inst_share"])
    for k, n, us, m in rows:
        wc = 4 * m["SQ_WAVE_CYCLES"]
        share = m["SQ_VALU_MFMA_BUSY_CYCLES"] / wc if wc else 0.0
        chip = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400.0) if us else 0.0
        ratio = m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"] if m["SQ_BUSY_CYCLES"] else 0.0
        wa = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0.0
        wi = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0.0
        w.writerow([k.split("(")[0][:100], n, f"{us:.1f}"] + [f"{m[c]:.4g}" for c in names] + [f"{share:.3f}", f"{chip:.3f}", f"{ratio:.2f}", f"{wa:.3f}", f"{wi:.3f}"])
        if m["SQ_INSTS_MFMA"] > 0:
            print(f"{k.split('(')[0][-60:]:60s} {n:3d} launches {us:7.1f} us  per-wave matrix share {share:5.3f}  chip matrix busy {chip:5.3f}  MFMA_BUSY/SQ_BUSY {ratio:5.2f}  parked {wa:4.2f}  issue-stalled {wi:4.2f}")
