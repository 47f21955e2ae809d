#!/bin/bash
# usage (GPU box): scratch/sweep_in_flight.sh   -- the development knobs again with two steps in flight (bench default), one box
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline > /tmp/g.json 2>/dev/null; python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/g.json"))
print(f"{sys.argv[1]:44s} {d['ms_per_step']:.4f} ms   one at a time {d['ms_per_step_one_in_flight']:.4f} ms", flush=True)
PY
}
run "default"
for spw in 3 4 6 8; do XEQ_WQ_STEPS_PER_WG=$spw run "steps per workgroup $spw"; done
XEQ_WQ_TAPER_FRAC=1.0 XEQ_WQ_TAPER_FRAC2=0.0 run "no taper"
XEQ_WQ_TAPER_FRAC=0.9 XEQ_WQ_TAPER_FRAC2=0.1 run "frac 0.9 / 0.1"
for w in 4 5 6 8; do XEQ_NODE_BLOCK_WAVES=$w run "node block waves $w"; done
run "default again"
