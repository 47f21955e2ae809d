#!/bin/bash
# round 6, GPU batch 19: K-split node block -- correctness at every size (forced on) and timing at MD sizes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{
echo "== node-block tests with the K-split form forced on"
XEQ_NODE_BLOCK_KSPLIT=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_nodeblock.py -x -q -m gpu 2>&1 | tail -5
echo "== launches (us), default policy"
for n in 21 192 1536 2048 4096; do timeout -k 10 300 python3 scratch/bench_nb2.py $n 2>&1 | tail -1; done
echo "== K-split forced at 4096"
XEQ_NODE_BLOCK_KSPLIT=1 timeout -k 10 300 python3 scratch/bench_nb2.py 4096 2>&1 | tail -1
echo "== whole steps"
python3 scratch/md_step.py 1 aspirin 2>&1 | tail -1
XEQ_NODE_BLOCK_KSPLIT=0 python3 scratch/md_step.py 1 aspirin 2>&1 | tail -1
python3 scratch/md_step.py 8 qm9 2>&1 | tail -1
XEQ_NODE_BLOCK_KSPLIT=0 python3 scratch/md_step.py 8 qm9 2>&1 | tail -1
python3 scratch/md_step.py 64 qm9 2>&1 | tail -1
XEQ_NODE_BLOCK_KSPLIT=0 python3 scratch/md_step.py 64 qm9 2>&1 | tail -1
python3 scratch/md_step.py 110 qm9 2>&1 | tail -1
XEQ_NODE_BLOCK_KSPLIT=0 python3 scratch/md_step.py 110 qm9 2>&1 | tail -1
} > $O/exp19.txt 2>&1
cat $O/exp19.txt
