#!/bin/bash
# usage (GPU box): scratch/ab_bench.sh <reps> <lib A> <lib B> [bench flags]   -- the same box, alternating: ms per step and the message kernels' per-launch times
reps=$1; A=$2; B=$3; shift 3
for r in $(seq $reps); do for L in $A $B; do
  XEQ_LIB_PATH=$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" > /tmp/ab.json 2> /tmp/ab.err
  python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["roofline"]["kernels_ms_per_step"]
print(f"{sys.argv[1].split('/')[-1]:28s} {d['ms_per_step']:.4f} ms  " + "  ".join(f"{n.replace('xeq_message_','')} {v*1e3:.1f}" for n, v in k.items()))
PY
done; done
