#!/bin/bash
# round 6, GPU experiment batch 2: node-block stage-boundary variants (3-stage ring without the LDS drain; boundary one tile later)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for n in intree nb_ring3 nb_late; do
  if [ $n = intree ]; then unset XEQ_LIB_PATH; else export XEQ_LIB_PATH=$V/libxeq_$n.so; fi
  echo "== $n"
  timeout -k 10 600 python3 -m pytest tests/test_gpu_nodeblock.py -x -q -m gpu 2>&1 | tail -3
  for rep in 1 2; do timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1; done
  timeout -k 10 300 python3 scratch/bench_nb2.py 86016 2>&1 | tail -1
  XEQ_NODE_BLOCK_WAVES=8 timeout -k 10 300 python3 scratch/bench_nb2.py 86016 2>&1 | tail -1 | sed 's/^/waves=8 /'
  XEQ_NODE_BLOCK_WAVES=4 timeout -k 10 300 python3 scratch/bench_nb2.py 18609 2>&1 | tail -1 | sed 's/^/waves=4 /'
done
} > $O/exp2.txt 2>&1
cat $O/exp2.txt
