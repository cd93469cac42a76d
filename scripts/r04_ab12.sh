#!/bin/bash
# planner constants under F16X2: per-FLOP efficiency of the 64-row tiles (no halo form) x slab cost of a K split, whole cfg2 step
for e in 0.65 0.5 0.4; do for sc in 9 7 5; do
  XV2_EFF64=$e XV2_SLAB_COST=$sc python bench.py --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eff64=$e slab=$sc', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04_ab12.log
done; done
