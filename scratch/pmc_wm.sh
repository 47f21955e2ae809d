#!/bin/bash
# usage (on the GPU box): scratch/pmc_wm.sh <tag>   -- SQ counters of the wm message kernels (bench_wm.py), three passes
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/scratch/bench_wm.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/scratch/bench_wm.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_c -- python3 $R/scratch/bench_wm.py > /dev/null 2>&1
cd $R
python3 scratch/pmc_sum.py gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b gpurun_out/pmc_${tag}_c > gpurun_out/pmc_${tag}.txt 2>&1
rm -rf gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b gpurun_out/pmc_${tag}_c
cat gpurun_out/pmc_${tag}.txt
