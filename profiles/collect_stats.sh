#!/bin/bash
# Run ON THE GPU BOX from the repo root:  profiles/collect_stats.sh <round-tag> [bench.py flags]
# Per-kernel time of the bench command: rocprofv3 --kernel-trace --stats (no counters in this pass).
# Leaves gpurun_out/<tag>_kernel_stats.csv (copied to profiles/ by hand after review) and the bench JSON line.
tag=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# library-GEMM selections are timed in an un-profiled run and replayed under the profiler (no tuning kernels in the trace)
python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --gemm-results $R/gpurun_out/gemm_$tag.csv "$@" > $R/gpurun_out/tune_$tag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-results $R/gpurun_out/gemm_$tag.csv "$@" > $R/gpurun_out/prof_$tag.log 2>&1
rc=$?
cd $R
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
grep '^{"metric"' gpurun_out/prof_$tag.log > gpurun_out/${tag}_bench_under_rocprof.json
rm -rf gpurun_out/prof_$tag   # raw traces are large; gpurun copies back at most 64 MiB
exit $rc
