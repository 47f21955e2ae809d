#!/bin/bash
# usage: scratch/build_wq_variant.sh <name> <source.hip> [extra flags]  -> scratch/variants/libxeq_<name>.so with that source as xeq_message_wq
name=$1; src=$2; shift 2
R=/root/repo; D=/tmp/var_$name; mkdir -p $D $R/scratch/variants
cp $src $D/xeq_message_wq.hip; cp $R/xequinet_amd/csrc/xeq_common.h $D/
sed -i 's#"../../include/xeq.h"#"/root/repo/include/xeq.h"#' $D/xeq_common.h
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c $D/xeq_message_wq.hip -o $D/xeq_message_wq.o -save-temps=obj 2>$D/err.txt || { cat $D/err.txt | head -30; echo "compile failed"; exit 1; }
awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{sp=$2} /\.wavefront_size:/{print substr(n,9,34), "vgpr",v,"spill",sp}' $D/*gfx950*.s | grep "wqILi11" | sed "s/^/$name /"
objs=""; for o in $R/xequinet_amd/csrc/build/*.o; do [ "$(basename $o)" = "xeq_message_wq.o" ] && objs="$objs $D/xeq_message_wq.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $objs
