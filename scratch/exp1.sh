#!/bin/bash
# round 6, GPU experiment batch 1 (run on the GPU box from the repo root): timing-only variants + L2 / LDS counters of the node block
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
V=$R/scratch/variants
{
echo "== message kernels: occupancy variants (bench_wq2.py; us per launch; ONLY_L builds run one l's units only)"
python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first"
for n in b0_w2 b0_w2s b0_w3 b1_w2 b1_w3 f0_w2 f0_w2s f0_w3 f1_w2 f1_w3; do
  XEQ_LIB_PATH=$V/libxeq_$n.so timeout -k 10 300 python3 scratch/bench_wq2.py 2>&1 | grep -E "general|first|Error|error" | head -4
done
echo "== node block: timing-only variants (bench_nb2.py; us per call)"
python3 scratch/bench_nb2.py 2>&1 | tail -1
for n in nb_nosave nb_saved0 nb_scratch0; do
  XEQ_LIB_PATH=$V/libxeq_$n.so timeout -k 10 300 python3 scratch/bench_nb2.py 2>&1 | tail -1
done
} > $O/exp1.txt 2>&1
cat $O/exp1.txt
cd /tmp && export TMPDIR=/tmp
n=0
run() { timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_nb6_$n -- python3 $R/scratch/bench_nb2.py > /dev/null 2>$O/pmc_nb6_$n.err; n=$((n+1)); }
run TCC_HIT_sum TCC_MISS_sum
run TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run TCC_REQ_sum TCC_READ_sum
run TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
run FETCH_SIZE
run WRITE_SIZE
cd $R
python3 scratch/pmc_sum.py $O/pmc_nb6_0 $O/pmc_nb6_1 $O/pmc_nb6_2 $O/pmc_nb6_3 $O/pmc_nb6_4 $O/pmc_nb6_5 $O/pmc_nb6_6 > $O/exp1_pmc_nb.txt 2>&1
rm -rf $O/pmc_nb6_?
cat $O/exp1_pmc_nb.txt
