#!/bin/bash
# build ablated variants of the igemm kernel (debug tool): scripts/ablate.sh 1 3 4 7 ...
cd $(dirname $0)/../xview2_amd
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DXV2_ABL=$a -x hip -c csrc/igemm_conv.hip -o build/igemm_abl$a.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libxv2_abl$a.so build/igemm_abl$a.o build/errors.o build/wgrad_conv.o build/norm_act.o build/pool.o build/pointwise.o build/loss_optim.o ) &
done
wait
ls -la libxv2_abl*.so
