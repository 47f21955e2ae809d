#!/bin/bash
# usage (GPU box): scratch/prof_py.sh <tag> <script.py> [args]   -- rocprofv3 kernel stats of a python script -> gpurun_out/<tag>_kstats.csv (+ printed)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kstats.csv && cut -d, -f1-6 gpurun_out/${tag}_kstats.csv | head -${PROF_HEAD:-8}
rm -rf gpurun_out/prof_$tag
