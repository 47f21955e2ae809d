#!/bin/bash
# round 6, GPU batch 26: the pair launch (update MLP beside dot_lin): tests of both fronts, MD latencies, aspirin trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_small_rows.py tests/test_gpu_mlp.py tests/test_gpu_interface.py -x -q -m gpu 2>&1 | tail -15 > $O/exp26_tests.txt || { cat $O/exp26_tests.txt; exit 1; }
tail -3 $O/exp26_tests.txt
timeout -k 10 600 python3 scratch/latency_md.py > $O/exp26_latency_on.txt 2>&1 || { tail -20 $O/exp26_latency_on.txt; exit 1; }
cut -c1-330 $O/exp26_latency_on.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "1 aspirin" "64 qm9"; do
  set -- $cfg
  tag=md_on_$1_$2
  python3 $R/scratch/md_step.py $1 $2 > $O/$tag.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq_$tag -- python3 $R/scratch/md_step.py $1 $2 > $O/seq_$tag.log 2>&1
  python3 $R/scratch/kernel_means.py $O/seq_$tag >> $O/$tag.txt
  rm -rf $O/seq_$tag
done
cd $R; cat $O/md_on_1_aspirin.txt; grep -E "replay|steady|TOTAL" $O/md_on_64_qm9.txt
