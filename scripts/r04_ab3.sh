#!/bin/bash
# planner: what a split-K plan is charged for its slab sum (XV2_SLAB_COST, K-tiles per block; default 3 f32x3 / 4 bf16)
for cfg in "resnest50 16" "resnet50 16" "resnet50 32"; do set -- $cfg
  echo "== $1 p$2  (XV2_SLAB_COST, img/s, ms)" | tee -a gpurun_out/r04_ab3.log
  for v in -1 6 10 16 1000 -1; do
    XV2_SLAB_COST=$v python bench.py --no-cpu-baseline --no-encoder-probe --no-prof --no-other-configs --steps 20 --warmup 5 --encoder $1 --precision $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04_ab3.log
  done
done
