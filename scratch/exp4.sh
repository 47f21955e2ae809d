#!/bin/bash
# round 6, GPU batch 4: phase stamps of the reverse message kernel (l = 0, 1, 2 waves), after the bf16 tail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
for L in 0 1 2; do
  echo "== stamps, reverse kernel, l = $L waves"
  XEQ_WQ_STAMPS=1 XEQ_LIB_PATH=$V/libxeq_stamps$L.so timeout -k 10 300 python3 scratch/bench_wq.py wq 2>&1 | grep -v "^wq\|amdgpu.ids" | tail -32
done
} > $O/exp4.txt 2>&1
cat $O/exp4.txt
