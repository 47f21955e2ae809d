#!/bin/bash
# usage: scratch/wq_regs.sh [extra hipcc flags]: VGPR / spill counts of the wq kernels per role (KS = 11)
mkdir -p /tmp/isa && cd /tmp/isa
for L in 0 1 2 all; do
  D="-DXEQ_WQ_ONLY_L=$L"; [ $L = all ] && D=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-function $D "$@" -c /root/repo/xequinet_amd/csrc/xeq_message_wq.hip -o wq_l$L.o -save-temps=obj 2>/dev/null
  awk '/^    \.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{sg=$2} /\.vgpr_spill_count:/{sp=$2} /\.wavefront_size:/{print substr(n,1,40), "vgpr",v,"sgpr",sg,"spill",sp}' xeq_message_wq-hip-amdgcn-amd-amdhsa-gfx950.s | grep "wqILi11" | sed "s/^/L=$L /"
  cp xeq_message_wq-hip-amdgcn-amd-amdhsa-gfx950.s wq_l$L.s
done
