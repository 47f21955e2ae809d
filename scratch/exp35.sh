#!/bin/bash
# round 6, GPU batch 35: update block + next message front with merged elementwise hand-overs (NodeChainSmall): tests, traces, latencies
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_small_rows.py tests/test_gpu_interface.py tests/test_gpu_mlp.py -x -q -m gpu 2>&1 | tail -15 > $O/exp35_tests.txt || { cat $O/exp35_tests.txt; exit 1; }
tail -3 $O/exp35_tests.txt
timeout -k 10 600 python3 scratch/latency_md.py 2>&1 | cut -c1-330 > $O/exp35_latency.txt; cat $O/exp35_latency.txt
cd /tmp && export TMPDIR=/tmp
python3 $R/scratch/md_step.py 1 aspirin > $O/md_on_1_aspirin.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq_x -- python3 $R/scratch/md_step.py 1 aspirin > $O/seq_x.log 2>&1
python3 $R/scratch/kernel_means.py $O/seq_x >> $O/md_on_1_aspirin.txt
rm -rf $O/seq_x
cd $R; grep -v amdgpu $O/md_on_1_aspirin.txt
