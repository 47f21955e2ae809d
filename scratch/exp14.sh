#!/bin/bash
# round 6, GPU batch 14: kernel trace of an MD-sized whole step (aspirin; 8 QM9 molecules; 64 QM9 molecules): kernel time vs gaps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "1 aspirin" "8 qm9" "64 qm9"; do
  set -- $cfg
  tag=md_$1_$2
  python3 $R/scratch/md_step.py $1 $2 > $O/$tag.txt 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/seq_$tag -- python3 $R/scratch/md_step.py $1 $2 > $O/seq_$tag.log 2>&1
  python3 - $O/seq_$tag $tag <<'PY' >> $O/$tag.txt
import csv, glob, re, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]
cuts = [i for i, r in enumerate(rows) if "k_load_padded_batch" in r["Kernel_Name"]] + [len(rows)]
steps = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
lens = collections.Counter(len(s) for s in steps)
L = lens.most_common(1)[0][0]
s = [x for x in steps if len(x) == L][-2]
t0 = int(s[0]["Start_Timestamp"]); t1 = int(s[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s)
print(f"one replayed step: {L} kernels, {(t1 - t0) / 1e3:.0f} us first start to last end, kernels busy {busy / 1e3:.0f} us, gaps {(t1 - t0 - busy) / 1e3:.0f} us")
for i, r in enumerate(s):
    print(f"{i + 1:3d}  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
PY
  rm -rf $O/seq_$tag
done
cd $R; head -80 $O/md_1_aspirin.txt; head -4 $O/md_8_qm9.txt $O/md_64_qm9.txt
