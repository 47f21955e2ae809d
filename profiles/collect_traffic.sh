#!/bin/bash
# Run ON THE GPU BOX from the repo root:  profiles/collect_traffic.sh <round-tag>
# HBM traffic of the xeq kernels of one bench.py evaluation: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots": they do not fit one pass), kernel trace only (no sys/hip traces).
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- python3 $R/bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --gemm-results $R/gpurun_out/gemm_$tag.csv > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $R
python3 profiles/summarise_traffic.py $tag gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
cp profiles/${tag}_traffic_pmc.csv profiles/traffic.json gpurun_out/   # only gpurun_out/ travels back
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE   # raw counter dumps exceed the 64 MiB copy-back limit
