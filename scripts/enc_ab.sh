#!/bin/bash
# usage (ON the GPU box): scripts/enc_ab.sh "ENV1=a ENV2=b" "ENV1=c" ... [-- bench args]
#   -> training-mode / eval-mode forward ms of the resnest50 encoder (bench.py --phase encoder-forward) per environment, 3 repetitions each
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
for rep in 1 2 3; do
for s in "${SETS[@]}"; do
  env $s python bench.py --phase encoder-forward --encoder resnest50 --steps 20 "$@" 2>/dev/null | TAG="$s" python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['encoder_forward']; print('%-40s train %.3f ms  mfma-kernels %.3f  eval %.3f ms' % (os.environ['TAG'], d['forward_ms'], d['mfma_kernels_ms'], d['eval_mode']['forward_ms']))"
done
done
