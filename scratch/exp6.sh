#!/bin/bash
# round 6, GPU batch 6: per-l sensitivity to the stream length (ONLY_L builds of the current kernels; us per launch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for len in 48 64 80 112 160; do
  for n in nf0 nf1 nf2 nb0 nb1 nb2; do
    echo -n "eps=$len $n: "
    XEQ_WQ_EDGES_PER_STREAM=$len XEQ_LIB_PATH=$V/libxeq_$n.so timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first" | tr '\n' ' '
    echo
  done
done
for len in 48 64 80 112; do
  echo -n "eps=$len all units: "
  XEQ_WQ_EDGES_PER_STREAM=$len timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first" | tr '\n' ' '
  echo
done
} > $O/exp6.txt 2>&1
cat $O/exp6.txt
