#!/bin/bash
# round 6, GPU batch 24: few-row forms of k_linear / k_mlp2 / k_update_uv: bit equality + the MD-sized steps with them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_small_rows.py tests/test_gpu_mlp.py tests/test_gpu_fullsize.py::test_whole_step_graph_replays_batches_of_changing_sizes -x -q -m gpu 2>&1 | tail -15 > $O/exp24_tests.txt || { cat $O/exp24_tests.txt; exit 1; }
cat $O/exp24_tests.txt
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export XEQ_SMALL_ROWS=0; else unset XEQ_SMALL_ROWS; fi
  for cfg in "1 aspirin" "64 qm9"; do
    set -- $cfg
    tag=md_${mode}_$1_$2
    python3 $R/scratch/md_step.py $1 $2 > $O/$tag.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq_$tag -- python3 $R/scratch/md_step.py $1 $2 > $O/seq_$tag.log 2>&1
    python3 $R/scratch/kernel_means.py $O/seq_$tag >> $O/$tag.txt
    rm -rf $O/seq_$tag
  done
done
cd $R; for f in $O/md_on_*.txt $O/md_off_*.txt; do echo "== $f"; grep -E "replay|k_linear|k_mlp2|k_update_uv|sb<|head_fused|TOTAL" $f; done
