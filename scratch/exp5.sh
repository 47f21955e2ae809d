#!/bin/bash
# round 6, GPU batch 5: full GPU suite on the current tree, reverse-kernel stamps, default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/exp5_tests.txt 2>&1
tail -5 $O/exp5_tests.txt
{
for L in 0 1 2; do
  echo "== stamps, reverse kernel, l = $L waves"
  XEQ_WQ_STAMPS=1 XEQ_LIB_PATH=$V/libxeq_stamps$L.so timeout -k 10 300 python3 scratch/bench_wq.py wq 2>&1 | grep -A 14 "reverse kernel"
done
} > $O/exp5_stamps.txt 2>&1
cat $O/exp5_stamps.txt
timeout -k 10 600 python3 bench.py > $O/exp5_bench.json 2> $O/exp5_bench.err
tail -c 1500 $O/exp5_bench.json
