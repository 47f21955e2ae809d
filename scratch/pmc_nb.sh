#!/bin/bash
# usage (on the GPU box): scratch/pmc_nb.sh <tag>   -- SQ counters of the node-block kernels (bench_nodeblock.py), passes of <= 8 SQ counters
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/scratch/bench_nodeblock.py quick > /dev/null 2>&1; n=$((n+1)); }
n=0
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS
run SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32
run SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_FLAT SQ_WAVES_LT_64 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
cd $R
python3 scratch/pmc_sum.py gpurun_out/pmc_${tag}_0 gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 > gpurun_out/pmc_${tag}.txt 2>&1
rm -rf gpurun_out/pmc_${tag}_?
cat gpurun_out/pmc_${tag}.txt
