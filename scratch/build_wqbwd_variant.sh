#!/bin/bash
# usage: scratch/build_wqbwd_variant.sh <name> [extra flags for the REVERSE half only]  -> scratch/variants/libxeq_<name>.so
name=$1; shift
R=/root/repo; D=/tmp/var_$name; rm -rf $D; mkdir -p $D $R/scratch/variants
base="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
( cd $R/xequinet_amd/csrc && /opt/rocm/bin/hipcc $base "$@" -c xeq_message_wq_bwd.hip -o $D/xeq_message_wq_bwd.o -save-temps=obj 2>$D/err.txt ) || { grep -v "not a recognized" $D/err.txt | head -20; echo "compile failed"; exit 1; }
python3 $R/scratch/kstats.py $D/*gfx950*.s "bwd_wq<3" | sed "s/^/$name /"
objs=""; for o in $R/xequinet_amd/csrc/build/*.o; do b=$(basename $o); if [ "$b" = "xeq_message_wq_bwd.o" ]; then objs="$objs $D/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $objs
