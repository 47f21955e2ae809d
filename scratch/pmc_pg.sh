#!/bin/bash
# usage (on the GPU box): scratch/pmc_pg.sh <tag>  -- counters of xeq_message_param_grad_mc (bench_param_grad.py): SQ activity, cache hits
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/scratch/bench_param_grad.py > /dev/null 2>&1; n=$((n+1)); }
n=0
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS
run SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
run TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
cd $R
python3 scratch/pmc_sum.py gpurun_out/pmc_${tag}_0 gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 > gpurun_out/pmc_${tag}.txt 2>&1
rm -rf gpurun_out/pmc_${tag}_?
grep -i "param_grad" -A40 gpurun_out/pmc_${tag}.txt | head -60
