#!/bin/bash
# Run ON the GPU box: the cfg2 fp32 step under several environments, round-robin, REPS times.   usage: scripts/ab_multi.sh REPS "ENV1" "ENV2" ...
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  for E in "$@"; do
    env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-prof --no-encoder-probe --no-split-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E', 'ms_per_step %.3f' % d['ms_per_step'], 'parity', (d.get('parity') or {}).get('pass'))"
  done
done
