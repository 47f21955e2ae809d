#!/bin/bash
# usage (GPU box, repo root): scratch/final_r06_a.sh   -- round 6, part A: bench lines, MD latencies, kernel statistics, step sequence
mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err || { tail -5 gpurun_out/r06_bench.err; exit 1; }
for w in md17_4096 qm9_8192; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_$w.json 2>/dev/null || exit 1; done
python bench.py --workload water_512 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r06_bench_water_512.json 2>/dev/null || exit 1
python scratch/latency_md.py > gpurun_out/r06_latency_md.txt 2>&1 || exit 1
profiles/collect_stats.sh r06 --in-flight 1 || exit 1                 # lone kernels: the durations the roofline line quotes
profiles/collect_stats.sh r06_in_flight2 || exit 1                    # the default command: two steps in flight, kernels side by side
bash scratch/step_sequence.sh r06 > /dev/null 2>&1
ls -la gpurun_out | tail -20
tail -c 600 gpurun_out/r06_bench.json
