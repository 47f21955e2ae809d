#!/bin/bash
# round 6, GPU batch 7: owners' rows one tile ahead in the reverse kernel (A/B against the bf16-tail build), then the full GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for rep in 1 2 3; do
  XEQ_LIB_PATH=$V/libxeq_bftail.so timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first" | tr '\n' ' '; echo
  timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first" | tr '\n' ' '; echo
done
bash scratch/ab_bench.sh 2 $V/libxeq_bftail.so $R/xequinet_amd/libxeq_hip.so
} > $O/exp7.txt 2>&1
cat $O/exp7.txt
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/exp7_tests.txt 2>&1
tail -5 $O/exp7_tests.txt
