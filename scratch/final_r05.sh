#!/bin/bash
# usage (GPU box, repo root): scratch/final_r05.sh   -- the round's bench lines, kernel statistics and the GPU suite, one box
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || exit 1
for w in md17_4096 qm9_8192; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_$w.json 2>/dev/null || exit 1; done
python bench.py --workload water_512 --steps 200 --warmup 20 > gpurun_out/r05_bench_water_512.json 2>/dev/null || exit 1
python scratch/latency_md.py > gpurun_out/r05_latency_md.txt 2>&1 || exit 1
profiles/collect_stats.sh r05 --in-flight 1 || exit 1                 # lone kernels: the durations the roofline line quotes
profiles/collect_stats.sh r05_in_flight2 || exit 1                    # the default command: two steps in flight, kernels side by side
python -m pytest tests -q -m gpu > gpurun_out/r05_gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gpu_tests.txt
tail -3 gpurun_out/r05_gpu_tests.txt
