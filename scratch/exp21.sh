#!/bin/bash
# round 6, GPU batch 21: reverse kernel with branch-free rows + hoisted chains as default (l = 0, 1, first-block l = 0); forward kernel without phase fences (+ deferred stores)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
bash scratch/ab_many.sh 3 $R/xequinet_amd/libxeq_hip.so $V/libxeq_ovl2.so $V/libxeq_fnofsb.so $V/libxeq_fdefnofsb.so
XEQ_LIB_PATH=$V/libxeq_ovl2.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wq or message or fused or first_block or model" 2>&1 | tail -3
} > $O/exp21.txt 2>&1
cat $O/exp21.txt
