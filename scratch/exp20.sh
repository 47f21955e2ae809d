#!/bin/bash
# round 6, GPU batch 20: reverse kernel, branch-free rows (S / E node gradients deferred) and the next pass's chains in front of the rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
bash scratch/ab_many.sh 3 $R/xequinet_amd/libxeq_hip.so $V/libxeq_dse.so $V/libxeq_ovl.so
XEQ_LIB_PATH=$V/libxeq_ovl.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wq or message or fused or first_block or model" 2>&1 | tail -3
} > $O/exp20.txt 2>&1
cat $O/exp20.txt
