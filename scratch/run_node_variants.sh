#!/bin/bash
# on the GPU box: per-kernel averages of scratch/bench_node.py for the listed variant libraries ("base" = the in-tree one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset XEQ_LIB_PATH; else export XEQ_LIB_PATH=$R/scratch/variants/libxeq_$v.so; fi
  rm -rf /tmp/nv_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nv_$v -o s -- python3 $R/scratch/bench_node.py > /tmp/nv_$v.log 2>&1
  f=$(find /tmp/nv_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys
v, f = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
out = []
for r in rows:
    n = r["Name"]
    if "xeq::" in n:
        out.append(f"{n.split('xeq::')[1].split('<')[0]} {float(r['AverageNs'])/1e3:.1f}")
print(v, "|", ", ".join(out))
PY
done
