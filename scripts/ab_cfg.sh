#!/bin/bash
# Run ON the GPU box: same-box A/B of another configuration's step under two environments.
# usage: scripts/ab_cfg.sh "ENV_A" "ENV_B" <bench args>     e.g. scripts/ab_cfg.sh XV2_SG_BF16=0 XV2_SG_BF16=1 --encoder resnest50 --precision 16
A=$1; B=$2; shift 2
for rep in 1 2; do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs --no-prof --no-encoder-probe --no-split-check "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.2f' % d['value'], 'loss %.5f' % d['loss'])"
  done
done
