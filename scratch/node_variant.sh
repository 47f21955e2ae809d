#!/bin/bash
# usage: scratch/node_variant.sh <name> <extra -D flags...>  -> scratch/variants/libxeq_<name>.so (xeq_node.hip rebuilt with the flags)
name=$1; shift
R=/root/repo; O=$R/xequinet_amd/csrc/build; mkdir -p $R/scratch/variants /tmp/nvar_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-function "$@" -c $R/xequinet_amd/csrc/xeq_node.hip -o /tmp/nvar_$name/node.o -save-temps=obj
awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{sp=$2} /\.wavefront_size:/{print substr(n,9,34), "vgpr",v,"spill",sp}' /tmp/nvar_$name/*gfx950*.s | grep "IfLi8\|norm_" | sed "s/^/$name /"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $O/xeq_graph.o $O/xeq_ops.o $O/xeq_message.o $O/xeq_message_mfma.o $O/xeq_message_sb.o $O/xeq_message_wm.o /tmp/nvar_$name/node.o
