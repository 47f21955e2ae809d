#!/bin/bash
# usage: scratch/kres.sh <file.hip> <name-pattern> [extra flags]  -> vgpr/sgpr/spill/lds per kernel
f=$1; pat=$2; shift 2
d=$(mktemp -d /tmp/kres.XXXX); cd $d
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -I/root/repo/include -c /root/repo/$f -o x.o -save-temps=cwd "$@" 2>/dev/null
awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.vgpr_spill_count:/{sp=$2} /\.group_segment_fixed_size:/{l=$2} /\.agpr_count:/{a=$2} /\.wavefront_size:/{print n, "vgpr",v,"agpr",a,"sgpr",s,"spill",sp,"lds",l}' *.s | grep -E "$pat" | sed 's/_ZN3xeq//'
echo $d
