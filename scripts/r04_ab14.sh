#!/bin/bash
# per-tap F16X2 kernels: prefetch depth 3 (two raw register sets) vs 4 (three) - 1x1 layers alone, then the whole step
for lib in libxv2.so libxv2_var_pf4.so; do
  echo "== $lib"
  XV2_LIB=/root/repo/xview2_amd/$lib XV2_SWEEP_H2=1 python scripts/sweep_tiles.py "1x1" 2>/dev/null | cut -c1-60
done
for lib in libxv2.so libxv2_var_pf4.so libxv2.so libxv2_var_pf4.so; do
  XV2_LIB=/root/repo/xview2_amd/$lib python bench.py --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done
