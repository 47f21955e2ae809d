#!/bin/bash
# round 6, GPU batch 13: node block, parameter vectors without memory access (timing only): what their ~80 just-in-time loads cost
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for rep in 1 2; do
  timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
  XEQ_LIB_PATH=$V/libxeq_nb_nopar.so timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
done
} > $O/exp13.txt 2>&1
cat $O/exp13.txt
