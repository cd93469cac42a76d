#!/bin/bash
for cfg in "resnest50 16" "resnest50 32"; do set -- $cfg
  echo "== $1 p$2  (XV2_FUSED_SPLAT, img/s, ms)" | tee -a gpurun_out/r04_ab4.log
  for v in 0 1 0 1; do
    XV2_FUSED_SPLAT=$v python bench.py --no-cpu-baseline --no-encoder-probe --no-prof --no-other-configs --steps 20 --warmup 5 --encoder $1 --precision $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04_ab4.log
  done
done
