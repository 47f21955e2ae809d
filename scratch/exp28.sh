#!/bin/bash
# round 6, GPU batch 28: stream length of the wq message kernels at MD sizes (water-64, water-512; LAMMPS-style replay)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
{
for sys in water64 water512; do
  for eps in 0 16 24 32 40 48 64 80; do
    if [ $eps = 0 ]; then unset XEQ_WQ_EDGES_PER_STREAM; else export XEQ_WQ_EDGES_PER_STREAM=$eps; fi
    echo -n "eps=$eps  "; timeout -k 10 120 python3 scratch/md_lmp.py $sys lmp 2>&1 | grep "ms per step"
  done
done
} > $O/exp28.txt 2>&1
cat $O/exp28.txt
