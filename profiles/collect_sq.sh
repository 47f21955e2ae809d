#!/bin/bash
# Run ON THE GPU BOX from the repo root:  profiles/collect_sq.sh <round-tag>
# SQ counters of the xeq kernels of bench.py's evaluation (eager, so every kernel is a launch of its own): matrix-pipe busy cycles,
# wave cycles and the wait buckets, one --pmc pass of eight SQ counters, kernel trace only (no sys/hip traces).
# -> gpurun_out/<tag>_sq_counters.csv (per kernel: launches and the mean of every counter per launch) and <tag>_sq_counters.txt
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $R/gpurun_out/sq_$tag -- python3 $R/bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/sq_$tag.log 2>&1
cd $R
python3 profiles/summarise_sq.py $tag gpurun_out/sq_$tag > gpurun_out/${tag}_sq_counters.txt
rm -rf gpurun_out/sq_$tag
cat gpurun_out/${tag}_sq_counters.txt
