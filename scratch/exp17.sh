#!/bin/bash
# round 6, GPU batch 17: node block with unconditional stores; reverse message kernel + pass-M stores deferred
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for rep in 1 2; do
  timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
  XEQ_LIB_PATH=$V/libxeq_nb_uncond.so timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
done
XEQ_LIB_PATH=$V/libxeq_nb_uncond.so timeout -k 10 600 python3 -m pytest tests/test_gpu_nodeblock.py -x -q -m gpu 2>&1 | tail -2
bash scratch/ab_many.sh 3 $V/libxeq_defer.so $V/libxeq_dfm.so
} > $O/exp17.txt 2>&1
cat $O/exp17.txt
