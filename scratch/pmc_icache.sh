#!/bin/bash
# usage (on the GPU box): scratch/pmc_icache.sh <tag>  -- instruction-fetch counters of the node-block forward kernel at 1 536 and 18 609 nodes
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_avail.txt 2>&1
grep -i -o "\b\(SQC\?_[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_LEVEL\)[A-Z_0-9]*\)\b" $R/gpurun_out/counters_avail.txt | sort -u > $R/gpurun_out/counters_ifetch.txt
n=0
run() { sz=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/scratch/bench_nodeblock.py $sz > $R/gpurun_out/pmc_${tag}_$n.log 2>&1; n=$((n+1)); }
for sz in small quick; do
  run $sz SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  run $sz SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS
done
cd $R
python3 scratch/pmc_sum.py gpurun_out/pmc_${tag}_0 gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 > gpurun_out/pmc_${tag}.txt 2>&1
tail -3 gpurun_out/pmc_${tag}_1.log >> gpurun_out/pmc_${tag}.txt
rm -rf gpurun_out/pmc_${tag}_?
cat gpurun_out/pmc_${tag}.txt
