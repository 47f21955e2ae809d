#!/bin/bash
# round 6, GPU batch 22: node block, unconditional stores pinned by scheduling fences
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for rep in 1 2; do
  timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
  XEQ_LIB_PATH=$V/libxeq_nb_pin.so timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
done
XEQ_LIB_PATH=$V/libxeq_nb_pin.so timeout -k 10 600 python3 -m pytest tests/test_gpu_nodeblock.py -x -q -m gpu 2>&1 | tail -2
} > $O/exp22.txt 2>&1
cat $O/exp22.txt
