#!/bin/bash
# round 6, GPU batch 9: next tile's records requested at the tile top (reverse kernel; l = 0 only / l = 0 and 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
bash scratch/ab_many.sh 3 $R/xequinet_amd/libxeq_hip.so $V/libxeq_reall1.so $V/libxeq_reall3.so
XEQ_LIB_PATH=$V/libxeq_reall3.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wq or message or fused or first_block or model" 2>&1 | tail -3
} > $O/exp9.txt 2>&1
cat $O/exp9.txt
