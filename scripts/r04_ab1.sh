#!/bin/bash
# round 4, first GPU call: gated-launch tests, then same-box A/B of XV2_COOP on the step and on the encoder forward
mkdir -p gpurun_out
XV2_TEST_WORKERS=0 timeout 900 python -m pytest tests/test_coop_gpu.py -x -q -m gpu > gpurun_out/r04_t_coop.log 2>&1
echo "coop tests rc=$?" | tee -a gpurun_out/r04_ab1.log
tail -5 gpurun_out/r04_t_coop.log
for enc in resnet50 resnest50; do for prec in 32 16; do
  echo "== $enc p$prec  (XV2_COOP value, img/s, ms)" | tee -a gpurun_out/r04_ab1.log
  scripts/ab_env.sh XV2_COOP 0 1 0 1 -- --encoder $enc --precision $prec --no-other-configs 2>&1 | tee -a gpurun_out/r04_ab1.log
done; done
for prec in 32 16; do for v in 0 1; do
  echo "== encoder-forward resnest50 p$prec XV2_COOP=$v" | tee -a gpurun_out/r04_ab1.log
  XV2_COOP=$v python bench.py --phase encoder-forward --encoder resnest50 --precision $prec --steps 30 --warmup 5 2>/dev/null | tail -1 | tee -a gpurun_out/r04_ab1.log
done; done
