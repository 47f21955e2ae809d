#!/bin/bash
# usage: scratch/build_variant.sh <name> <source stem, e.g. xeq_linear> [extra flags]  -> scratch/variants/libxeq_<name>.so (csrc/build.py's flags)
name=$1; stem=$2; shift; shift
R=/root/repo; D=/tmp/var_$name; rm -rf $D; mkdir -p $D $R/scratch/variants
base="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
( cd $R/xequinet_amd/csrc && /opt/rocm/bin/hipcc $base "$@" -c $stem.hip -o $D/$stem.o -save-temps=obj 2>$D/err.txt ) || { grep -v "not a recognized" $D/err.txt | head -20; echo "compile failed"; exit 1; }
python3 $R/scratch/kstats.py $D/*gfx950*.s "${KFILTER:-k_}" | sed "s/^/$name /"
if [ -z "$NOLINK" ]; then
objs=""; for o in $R/xequinet_amd/csrc/build/*.o; do b=$(basename $o); if [ "$b" = "$stem.o" ]; then objs="$objs $D/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libxeq_$name.so $objs
fi
