#!/bin/bash
# usage: scratch/build_variant.sh <name> <extra -D flags...>   -> scratch/variants/libxeq_<name>.so (wm file rebuilt with the flags)
name=$1; shift
R=/root/repo; O=$R/xequinet_amd/csrc/build; mkdir -p $R/scratch/variants /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-function "$@" -c $R/xequinet_amd/csrc/xeq_message_wm.hip -o /tmp/var_$name/wm.o -save-temps=obj 2>/dev/null
awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{sp=$2} /\.wavefront_size:/{print substr(n,9,24), "vgpr",v,"spill",sp}' /tmp/var_$name/*gfx950*.s | grep Li11E | sed "s/^/$name /"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $O/xeq_graph.o $O/xeq_ops.o $O/xeq_message.o $O/xeq_message_mfma.o $O/xeq_message_sb.o /tmp/var_$name/wm.o $O/xeq_node.o
