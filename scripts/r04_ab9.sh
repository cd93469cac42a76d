#!/bin/bash
XV2_TEST_WORKERS=0 python -m pytest tests/test_conv_shapes_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
XV2_TEST_WORKERS=0 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "resnet50 and not resnest" 2>&1 | tail -3
for v in 0 1 0 1; do
  XV2_STEM7W=$v python bench.py --no-cpu-baseline --no-encoder-probe --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stem7w=$v', d['value'], d['ms_per_step'], d['loss'], d['parity']['pass'] if 'parity' in d else None, [ (r['kernel'], r['tflops'], r['ms_per_step']) for r in d['roofline']['per_kernel'] if 'rgb' in r['kernel']])" | tee -a gpurun_out/r04_ab9.log
done
