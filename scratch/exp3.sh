#!/bin/bash
# round 6, GPU batch 3: bf16 tail of the filter + no selects on invalid rows: correctness (message tests) and A/B against the round-5 kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wq or message or fused or first_block or model" 2>&1 | tail -5
for rep in 1 2 3; do
  XEQ_LIB_PATH=$V/libxeq_wqbase.so timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first"
  timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first"
done
bash scratch/ab_bench.sh 2 $V/libxeq_wqbase.so $R/xequinet_amd/libxeq_hip.so
} > $O/exp3.txt 2>&1
cat $O/exp3.txt
