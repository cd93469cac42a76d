#!/bin/bash
XV2_TEST_WORKERS=0 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "resnest or split_attention or splat or gate_and" 2>&1 | tail -3
A="--encoder resnest200 --type post --dmg_model fused --attention --ppm --deep_supervision --precision 16 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 10 --warmup 4"
for v in 0 1 0 1; do
  XV2_SPLAT_TAIL=$v python bench.py $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 tail=$v', d['value'], d['ms_per_step'], d['loss'])" | tee -a gpurun_out/r04_ab7.log
done
for v in 0 1 0 1; do
  XV2_SPLAT_TAIL=$v python bench.py --encoder resnest50 --precision 16 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 tail=$v', d['value'], d['ms_per_step'], d['loss'])" | tee -a gpurun_out/r04_ab7.log
done
python scripts/host_time.py --encoder resnest200 --type post --dmg_model fused --precision 16 --attention --ppm --deep_supervision 2>&1 | tail -2
