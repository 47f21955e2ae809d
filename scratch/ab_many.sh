#!/bin/bash
# usage (GPU box): scratch/ab_many.sh <reps> <lib> [<lib> ...]  -- the same box, round-robin: ms per step and the message kernels' per-launch times
reps=$1; shift
for r in $(seq $reps); do for L in "$@"; do
  XEQ_LIB_PATH=$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err
  python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["roofline"]["kernels_ms_per_step"]
print(f"{sys.argv[1].split('/')[-1]:34s} {d['ms_per_step']:.4f} ms  " + "  ".join(f"{n.replace('xeq_message_','')} {v*1e3:.1f}" for n, v in k.items()))
PY
done; done
