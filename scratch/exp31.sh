#!/bin/bash
# round 6, GPU batch 31: where the few-row forms stop paying: whole-step replay at 1.2 k .. 5.9 k atoms with the forms off / on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{
for nm in 64 100 128 160 200 256 320; do
  for lim in 0 100000; do
    echo -n "XEQ_SMALL_ROWS=$lim  "; XEQ_SMALL_ROWS=$lim XEQ_NODE_BLOCK=0 timeout -k 10 200 python3 scratch/md_step.py $nm qm9 2>&1 | grep replay
  done
  echo -n "node block        "; XEQ_SMALL_ROWS=0 XEQ_NODE_BLOCK_MIN_NODES=1 timeout -k 10 200 python3 scratch/md_step.py $nm qm9 2>&1 | grep replay
done
} > $O/exp31.txt 2>&1
cat $O/exp31.txt
