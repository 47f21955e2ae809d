#!/bin/bash
# usage: scratch/build_variant.sh <name> <file.hip> <extra flags...>  -> scratch/variants/libxeq_<name>.so (that file rebuilt with the flags)
name=$1; src=$2; shift 2
R=/root/repo; O=$R/xequinet_amd/csrc/build; mkdir -p $R/scratch/variants /tmp/var_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-function "$@" -c $R/xequinet_amd/csrc/$src -o /tmp/var_$name/$base.o -save-temps=obj 2>/dev/null || { echo "compile failed"; exit 1; }
awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{sp=$2} /\.wavefront_size:/{print substr(n,9,30), "vgpr",v,"spill",sp}' /tmp/var_$name/*gfx950*.s | grep -v rocprim | grep "Li11E\|k_[a-z_]*E" | sed "s/^/$name /"
objs=""; for o in $O/*.o; do [ "$(basename $o)" = "$base.o" ] && objs="$objs /tmp/var_$name/$base.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $objs
