#!/bin/bash
# where the time of a 1x1 layer goes (per-tap F16X2 kernel): ablated builds (scripts/ablate.sh 1 2 8 9) on l3.conv3 / l3.conv1 / l2.conv3
for lib in libxv2.so libxv2_var_abl1.so libxv2_var_abl2.so libxv2_var_abl8.so libxv2_var_abl9.so; do
  echo "== $lib"
  XV2_LIB=/root/repo/xview2_amd/$lib XV2_SWEEP_H2=1 python scripts/sweep_tiles.py "l3.conv3" "l3.conv1" "l2.conv3" 2>/dev/null | cut -c1-75
done
