#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "water512 lmp" "water64 lmp"; do
  set -- $cfg
  tag=mdk_$1_$2
  rocprofv3 --kernel-trace --output-format csv -d $O/seq_$tag -- python3 $R/scratch/md_lmp.py $1 $2 > $O/$tag.txt 2>&1
  python3 $R/scratch/kernel_means_all.py $O/seq_$tag 40 >> $O/$tag.txt
  rm -rf $O/seq_$tag
  echo "== $tag"; grep -v "amdgpu.ids\|rocprofv3\|Opened\|HSA version" $O/$tag.txt
done
