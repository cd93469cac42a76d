#!/bin/bash
A="--encoder resnest200 --type post --dmg_model fused --attention --ppm --deep_supervision --precision 16 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 10 --warmup 4"
for g in "" "--graph" "" "--graph"; do
  python bench.py $A $g 2> gpurun_out/r04_ab6.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 [$g]', d['value'], d['ms_per_step'], d['launch'], d['loss'])" | tee -a gpurun_out/r04_ab6.log
  tail -2 gpurun_out/r04_ab6.err | cut -c1-300
done
B="--encoder resnest50 --precision 16 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 20 --warmup 5"
for g in "" "--graph" "" "--graph"; do
  python bench.py $B $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 [$g]', d['value'], d['ms_per_step'], d['launch'], d['loss'])" | tee -a gpurun_out/r04_ab6.log
done
