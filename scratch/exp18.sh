#!/bin/bash
# round 6, GPU batch 18: the FUSED node block at MD sizes (what a K-split small-N form would start from)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{
for n in 21 192 1536 4096; do timeout -k 10 300 python3 scratch/bench_nb2.py $n 2>&1 | tail -1; done
python3 scratch/md_step.py 1 aspirin 2>&1 | tail -1
XEQ_NODE_BLOCK_MIN_NODES=0 python3 scratch/md_step.py 1 aspirin 2>&1 | tail -1
python3 scratch/md_step.py 64 qm9 2>&1 | tail -1
XEQ_NODE_BLOCK_MIN_NODES=0 python3 scratch/md_step.py 64 qm9 2>&1 | tail -1
} > $O/exp18.txt 2>&1
cat $O/exp18.txt
