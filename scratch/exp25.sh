#!/bin/bash
# round 6, GPU batch 25: few-row forms: tests, MD latencies (on / off), step trace of water-64 and water-512
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_small_rows.py tests/test_gpu_mlp.py tests/test_gpu_fullsize.py::test_whole_step_graph_replays_batches_of_changing_sizes -x -q -m gpu 2>&1 | tail -15 > $O/exp25_tests.txt || { cat $O/exp25_tests.txt; exit 1; }
tail -3 $O/exp25_tests.txt
timeout -k 10 600 python3 scratch/latency_md.py > $O/exp25_latency_on.txt 2>&1 || { tail -20 $O/exp25_latency_on.txt; exit 1; }
XEQ_SMALL_ROWS=0 timeout -k 10 600 python3 scratch/latency_md.py > $O/exp25_latency_off.txt 2>&1 || exit 1
echo "== on"; cut -c1-330 $O/exp25_latency_on.txt; echo "== off"; cut -c1-330 $O/exp25_latency_off.txt
