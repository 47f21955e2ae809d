#!/bin/bash
# round 6, GPU batch 33: plan-free second graph for replays on an unchanged list (LAMMPS-style front)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_interface.py -x -q -m gpu 2>&1 | tail -4 > $O/exp33_tests.txt || { cat $O/exp33_tests.txt; exit 1; }
cat $O/exp33_tests.txt
timeout -k 10 600 python3 scratch/latency_md.py 2>&1 | cut -c1-200 > $O/exp33_latency.txt; cat $O/exp33_latency.txt
