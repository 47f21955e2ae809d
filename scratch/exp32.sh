#!/bin/bash
# round 6, GPU batch 32: the walk plan built on a side stream beside reverse map / edge vectors / first-block front (whole-step graph)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "whole_step or in_flight or capacity or replays" 2>&1 | tail -4 > $O/exp32_tests.txt || { cat $O/exp32_tests.txt; exit 1; }
cat $O/exp32_tests.txt
{
for rep in 1 2 3; do
  for f in 0 1; do
    echo -n "XEQ_FORK_PLAN=$f one at a time: "; XEQ_FORK_PLAN=$f timeout -k 10 300 python3 bench.py --in-flight 1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
for rep in 1 2; do
  for f in 0 1; do
    echo -n "XEQ_FORK_PLAN=$f two in flight: "; XEQ_FORK_PLAN=$f timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
} > $O/exp32.txt 2>&1
cat $O/exp32.txt
