#!/bin/bash
# build a variant of the library with extra compiler flags for ONE source (A/B runs on the same box):
#   scripts/variant.sh NAME SOURCE.hip "-DXV2_PF=3 ..."   ->  xview2_amd/abl/xv2_NAME.so   (use with XV2_LIB=...)
cd $(dirname $0)/../xview2_amd
NAME=$1; SRC=$2; FLAGS=$3
BASE=$(basename $SRC .hip)
mkdir -p abl
OBJS=$(ls build/*.o | grep -v "build/$BASE.o\|igemm_abl\|_var_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -x hip -c csrc/$SRC -o build/${BASE}_var_$NAME.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/xv2_$NAME.so build/${BASE}_var_$NAME.o $OBJS && ls -la abl/xv2_$NAME.so
