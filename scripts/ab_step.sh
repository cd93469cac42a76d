#!/bin/bash
# Run ON the GPU box: same-box A/B of the cfg2 fp32 step (and the resnest50 encoder forward) under two environments.
# usage: scripts/ab_step.sh "ENV_A" "ENV_B" [bench args]       e.g.  scripts/ab_step.sh "XV2_SG=0" "XV2_SG=1"
A=$1; B=$2; shift 2
for rep in 1 2; do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-prof "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ef=d.get('encoder_forward') or []
print('$E', 'ms_per_step %.3f' % d['ms_per_step'], 'img/s %.2f' % d['value'], 'parity', (d.get('parity') or {}).get('pass'), ' | '.join('enc p%s %.3f ms' % (e['precision'], e['forward_ms']) for e in ef))"
  done
done
