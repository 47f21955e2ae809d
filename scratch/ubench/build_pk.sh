#!/bin/bash
# builds scratch/ubench/pk_after_mfma (gfx950; runs on the GPU box): the same kernel with and without packed-fp32 instructions
cd $(dirname $0)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -DBUILD_PACKED -c pk_after_mfma.hip -o /tmp/pk_p.o &&
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -DBUILD_MAIN -Xclang -target-feature -Xclang -packed-fp32-ops -c pk_after_mfma.hip -o /tmp/pk_s.o 2>&1 | grep -v "not a recognized feature" ;
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/pk_p.o /tmp/pk_s.o -o pk_after_mfma.bin
