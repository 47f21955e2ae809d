#!/bin/bash
# round 6, GPU batch 8: two stream classes (48-edge table streams, l = 0 units three at a time): message tests, A/B, sweep of the class geometry
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
one() {  # label, env...
  lab=$1; shift
  env "$@" python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err
  python3 - "$lab" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab.json")); k = d["roofline"]["kernels_ms_per_step"]
    print(f"{sys.argv[1]:44s} {d['ms_per_step']:.4f} ms (one at a time {d['ms_per_step_one_in_flight']:.4f})  " + "  ".join(f"{n.replace('xeq_message_','')} {v*1e3:.1f}" for n, v in k.items()))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab.err").read()[-300:])
PY
}
{
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wq or message or fused or first_block or model" 2>&1 | tail -3
for rep in 1 2; do
  one "baseline (owners ahead, one class, 80)" XEQ_LIB_PATH=$V/libxeq_pball.so
  one "classes 48 x3 (default)" XEQ_A=1
done
one "classes 48 x2" XEQ_WQ_LONG_MULT=2
one "classes 48 x4" XEQ_WQ_LONG_MULT=4
one "classes 40 x3" XEQ_WQ_EDGES_PER_STREAM=40 XEQ_WQ_LONG_MULT=3
one "classes 40 x4" XEQ_WQ_EDGES_PER_STREAM=40 XEQ_WQ_LONG_MULT=4
one "classes 56 x3" XEQ_WQ_EDGES_PER_STREAM=56 XEQ_WQ_LONG_MULT=3
one "classes 64 x2" XEQ_WQ_EDGES_PER_STREAM=64 XEQ_WQ_LONG_MULT=2
one "one class 48" XEQ_WQ_EDGES_PER_STREAM=48 XEQ_WQ_LONG_MULT=1
one "one class 80" XEQ_WQ_EDGES_PER_STREAM=80 XEQ_WQ_LONG_MULT=1
one "classes 48 x3, 1 unit per wg" XEQ_WQ_STEPS_PER_WG=1
one "classes 48 x3, 3 units per wg" XEQ_WQ_STEPS_PER_WG=3
} > $O/exp8.txt 2>&1
cat $O/exp8.txt
