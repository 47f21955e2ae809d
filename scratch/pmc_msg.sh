#!/bin/bash
# usage (on the GPU box): scratch/pmc_msg.sh <tag>   -- SQ counters of the message kernels (bench_msg.py), two passes
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/scratch/bench_msg.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/scratch/bench_msg.py > /dev/null 2>&1
cd $R
python3 scratch/pmc_sum.py gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b
