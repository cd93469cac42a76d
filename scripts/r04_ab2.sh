#!/bin/bash
# fused BatchNorm-backward statistics in the consumer's backward-data epilogue: tests, then same-box A/B per size threshold
mkdir -p gpurun_out
XV2_TEST_WORKERS=0 timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bn_backward_statistics" 2>&1 | tail -4
for cfg in "resnet50 32" "resnet50 16" "resnest50 16"; do set -- $cfg
  echo "== $1 p$2  (XV2_FUSE_BN_BWD / MAX, img/s, ms)" | tee -a gpurun_out/r04_ab2.log
  for v in "0 1e12" "1 1e12" "1 9e6" "1 2.2e6" "0 1e12" "1 1e12"; do set -- $cfg $v
    XV2_FUSE_BN_BWD=$3 XV2_FUSE_BN_BWD_MAX=$4 python bench.py --no-cpu-baseline --no-encoder-probe --no-prof --no-other-configs --steps 20 --warmup 5 --encoder $1 --precision $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 $4', d['value'], d['ms_per_step'], d['loss'])" | tee -a gpurun_out/r04_ab2.log
  done
done
