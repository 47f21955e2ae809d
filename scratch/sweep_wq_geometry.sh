#!/bin/bash
# usage (GPU box): scratch/sweep_wq_geometry.sh   -- chunk geometry of the wq message kernels (development env knobs), one box
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline > /tmp/g.json 2>/dev/null; python - "$1" <<'PY'
import json, sys
d = json.load(open("/tmp/g.json")); k = d["roofline"]["kernels_ms_per_step"]
print(f"{sys.argv[1]:44s} {d['ms_per_step']:.4f} ms  " + "  ".join(f"{n.replace('xeq_message_','')} {v*1e3:.1f}" for n, v in k.items()))
PY
}
run "default (steps/wg auto, 0.8 / 0.2 / div 3)"
for spw in 3 4 6 8; do XEQ_WQ_STEPS_PER_WG=$spw run "steps per workgroup $spw"; done
XEQ_WQ_TAPER_FRAC=0.7 run "frac 0.7"
XEQ_WQ_TAPER_FRAC=0.6 XEQ_WQ_TAPER_FRAC2=0.3 run "frac 0.6 / 0.3"
XEQ_WQ_TAPER_FRAC=0.9 XEQ_WQ_TAPER_FRAC2=0.1 run "frac 0.9 / 0.1"
XEQ_WQ_TAPER_DIV=2 run "div 2"
XEQ_WQ_TAPER_DIV=4 run "div 4"
XEQ_WQ_TAPER_FRAC=1.0 XEQ_WQ_TAPER_FRAC2=0.0 run "no taper"
run "default again"
