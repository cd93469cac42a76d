#!/bin/bash
# runtime knobs: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG)
for cfg in "resnest50 16" "resnet50 32" "resnet50 16"; do set -- $cfg
  echo "== $1 p$2  (HIP_FORCE_DEV_KERNARG, img/s, ms)" | tee -a gpurun_out/r04_ab5.log
  for v in 0 1 0 1; do
    HIP_FORCE_DEV_KERNARG=$v python bench.py --no-cpu-baseline --no-encoder-probe --no-prof --no-other-configs --steps 20 --warmup 5 --encoder $1 --precision $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r04_ab5.log
  done
done
env | grep -i "HIP_\|HSA_\|ROC" | head
