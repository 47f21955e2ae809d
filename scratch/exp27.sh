#!/bin/bash
# round 6, GPU batch 27: kernels of the LAMMPS-style replayed step at the three MD sizes, and of the GROMACS-style whole-step graph
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "aspirin lmp" "water64 lmp" "water512 lmp" "aspirin gmx" "water512 gmx"; do
  set -- $cfg
  tag=mdk_$1_$2
  rocprofv3 --kernel-trace --output-format csv -d $O/seq_$tag -- python3 $R/scratch/md_lmp.py $1 $2 > $O/$tag.txt 2>&1
  python3 $R/scratch/kernel_means_all.py $O/seq_$tag 40 >> $O/$tag.txt
  rm -rf $O/seq_$tag
  echo "== $tag"; grep -v amdgpu.ids $O/$tag.txt
done
