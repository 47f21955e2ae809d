#!/bin/bash
# round 6, GPU batch 29: XPainnUpdate.forward as one launch per 16-node tile: tests, aspirin / qm9-64 step traces
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_small_rows.py tests/test_gpu_mlp.py -x -q -m gpu 2>&1 | tail -15 > $O/exp29_tests.txt || { cat $O/exp29_tests.txt; exit 1; }
tail -3 $O/exp29_tests.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "1 aspirin" "64 qm9"; do
  set -- $cfg
  tag=md_on_$1_$2
  python3 $R/scratch/md_step.py $1 $2 > $O/$tag.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq_$tag -- python3 $R/scratch/md_step.py $1 $2 > $O/seq_$tag.log 2>&1
  python3 $R/scratch/kernel_means.py $O/seq_$tag >> $O/$tag.txt
  rm -rf $O/seq_$tag
done
cd $R; grep -v amdgpu $O/md_on_1_aspirin.txt; grep -E "replay|steady|TOTAL|update_block|uv_fwd" $O/md_on_64_qm9.txt
