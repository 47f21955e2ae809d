#!/bin/bash
# round 6, GPU batch 16: + forward kernel's node stores deferred to the next tile's top (l = 0, 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
bash scratch/ab_many.sh 3 $V/libxeq_defer.so $V/libxeq_dfall.so
XEQ_LIB_PATH=$V/libxeq_dfall.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wq or message or fused or first_block or model" 2>&1 | tail -3
} > $O/exp16.txt 2>&1
cat $O/exp16.txt
