#!/bin/bash
for prec in 32 16; do
for v in 0 1 0 1; do
  XV2_THIN_CT=$v python bench.py --precision $prec --no-cpu-baseline --no-encoder-probe --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p$prec thin_ct=$v', d['value'], d['ms_per_step'], d['loss'], [ (r['kernel'], r['tflops'], r['ms_per_step']) for r in d['roofline']['per_kernel'] if 'convT' in r['kernel'] or '128,32,4,1' in r['kernel']])" | tee -a gpurun_out/r04_ab10.log
done; done
