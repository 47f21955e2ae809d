#!/bin/bash
# usage: scratch/build_wq_variant.sh <name> <source.hip> [extra flags]  -> scratch/variants/libxeq_<name>.so with that source as
# xeq_message_wq.hip: both halves (forward: max-ILP scheduler; reverse, -DXEQ_WQ_PART_BWD: default scheduler), csrc/build.py's flags
name=$1; src=$2; shift 2
R=/root/repo; D=/tmp/var_$name; rm -rf $D; mkdir -p $D/f $D/b $R/scratch/variants
cp $src $D/xeq_message_wq.hip; cp $R/xequinet_amd/csrc/xeq_common.h $D/
sed -i 's#"../../include/xeq.h"#"/root/repo/include/xeq.h"#' $D/xeq_common.h
printf '#define XEQ_WQ_PART_BWD 1\n#include "../xeq_message_wq.hip"\n' > $D/b/xeq_message_wq_bwd.hip
cp $D/xeq_message_wq.hip $D/f/xeq_message_wq.hip; cp $D/xeq_common.h $D/f/
base="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function"
( cd $D/f && /opt/rocm/bin/hipcc $base -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c xeq_message_wq.hip -o $D/xeq_message_wq.o -save-temps=cwd 2>$D/err.txt || { head -30 $D/err.txt; echo "compile failed (forward half)"; } ) &
( cd $D/b && /opt/rocm/bin/hipcc $base "$@" -c xeq_message_wq_bwd.hip -o $D/xeq_message_wq_bwd.o -save-temps=cwd 2>$D/err_b.txt || { head -30 $D/err_b.txt; echo "compile failed (reverse half)"; } ) &
wait
[ -f $D/xeq_message_wq.o ] && [ -f $D/xeq_message_wq_bwd.o ] || exit 1
awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{sp=$2} /\.wavefront_size:/{print substr(n,9,34), "vgpr",v,"spill",sp}' $D/f/*gfx950*.s $D/b/*gfx950*.s | grep "wqILi3E\|wqILi11E" | sed "s/^/$name /"
objs=""; for o in $R/xequinet_amd/csrc/build/*.o; do b=$(basename $o); if [ "$b" = "xeq_message_wq.o" ] || [ "$b" = "xeq_message_wq_bwd.o" ]; then objs="$objs $D/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $objs
