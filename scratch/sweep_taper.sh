#!/bin/bash
# bench.py under a few chunk-taper settings of the wq message kernels (development env switches of wq_geometry)
cd ${GRAFT_REPO_ROOT:-.}
run() { echo -n "$1 : "; env $1 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['roofline']['kernels_ms_per_step'].items()})"; }
run "XEQ_NOP=1"
run "XEQ_WQ_TAPER_FRAC=0.7"
run "XEQ_WQ_TAPER_FRAC=0.9 XEQ_WQ_TAPER_FRAC2=0.1"
run "XEQ_WQ_TAPER_DIV=2"
run "XEQ_WQ_TAPER_DIV=4"
run "XEQ_WQ_STEPS_PER_WG=4"
run "XEQ_WQ_STEPS_PER_WG=6"
run "XEQ_NOP=2"
