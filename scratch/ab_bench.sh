#!/bin/bash
# usage (GPU box): scratch/ab_bench.sh <variant lib name> [rounds]  -- bench.py step time, in-tree library vs scratch/variants/libxeq_<name>.so, alternating
v=$1; n=${2:-3}
for i in $(seq $n); do
  a=$(python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  b=$(XEQ_LIB_PATH=$GRAFT_REPO_ROOT/scratch/variants/libxeq_$v.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "round $i: in-tree $a ms   $v $b ms"
done
