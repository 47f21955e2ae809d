#!/bin/bash
# usage (GPU box, repo root): scratch/final_r06_b.sh   -- round 6, part B: PMC traffic, SQ counters, the 65k batch on one card, the GPU suite
mkdir -p gpurun_out
bash profiles/collect_traffic.sh r06 > gpurun_out/r06_traffic.log 2>&1
bash profiles/collect_sq.sh r06 > gpurun_out/r06_sq.log 2>&1
python bench.py --workload qm9_65536 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_qm9_65536_g1.json 2>/dev/null
python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gpu_tests.txt
tail -3 gpurun_out/r06_gpu_tests.txt
ls -la gpurun_out | tail -12
