#!/bin/bash
# round 6, GPU batch 10: stamps of the current message kernels (reverse l = 0, 1; forward l = 0), then the full GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for L in 0 1; do
  echo "== stamps, reverse kernel, l = $L waves"
  XEQ_WQ_STAMPS=1 XEQ_LIB_PATH=$V/libxeq_stamps$L.so timeout -k 10 300 python3 scratch/bench_wq.py wq 2>&1 | grep -A 14 "reverse kernel"
done
echo "== stamps, forward kernel, l = 0 waves"
XEQ_WQ_STAMPS=1 XEQ_LIB_PATH=$V/libxeq_stampsf.so timeout -k 10 300 python3 scratch/bench_wq.py wq 2>&1 | grep -A 12 "forward kernel"
} > $O/exp10_stamps.txt 2>&1
cat $O/exp10_stamps.txt
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/exp10_tests.txt 2>&1
tail -5 $O/exp10_tests.txt
