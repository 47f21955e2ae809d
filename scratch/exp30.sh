#!/bin/bash
# round 6, GPU batch 30: sb reverse kernel with the next edge's gathers requested ahead: bits against the previous build, timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
XEQ_LIB_PATH=$R/scratch/variants/libxeq_prev.so timeout -k 10 200 python3 scratch/sb_ab.py /tmp/sb_prev.pt || exit 1
timeout -k 10 200 python3 scratch/sb_ab.py /tmp/sb_new.pt || exit 1
python3 scratch/sb_ab.py --compare /tmp/sb_prev.pt /tmp/sb_new.pt > $O/exp30_bits.txt 2>&1; cat $O/exp30_bits.txt
for rep in 1 2; do
  XEQ_LIB_PATH=$R/scratch/variants/libxeq_prev.so timeout -k 10 200 python3 scratch/md_step.py 1 aspirin 2>&1 | grep replay
  timeout -k 10 200 python3 scratch/md_step.py 1 aspirin 2>&1 | grep replay
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/seq_x -- python3 $R/scratch/md_step.py 1 aspirin > $O/seq_x.log 2>&1
python3 $R/scratch/kernel_means.py $O/seq_x | grep -E "sb<|TOTAL"
rm -rf $O/seq_x
