#!/bin/bash
# build ablated variants of the igemm kernel (debug tool): scripts/ablate.sh 1 8 9 ...   (XV2_ABL bits: 1 no global loads,
# 2 no LDS stores, 4 no MFMA (fp32 forms only), 8 no epilogue); use with XV2_LIB=xview2_amd/libxv2_abl<N>.so
cd $(dirname $0)/../xview2_amd
OBJS=$(ls build/*.o | grep -v "igemm_conv.o\|igemm_abl\|_var_")
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DXV2_ABL=$a -x hip -c csrc/igemm_conv.hip -o build/igemm_abl$a.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libxv2_abl$a.so build/igemm_abl$a.o $OBJS ) &
done
wait
ls -la libxv2_abl*.so
