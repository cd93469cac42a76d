#!/bin/bash
# Run ON the GPU box: a --precision 16 step (extra bench args in $ARGS) under several environments, round-robin.   usage: ARGS="--encoder resnest50" scripts/ab_multi16.sh REPS "ENV1" ...
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  for E in "$@"; do
    env $E python bench.py --precision 16 $ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-prof --no-encoder-probe --no-split-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$E', 'ms_per_step %.3f' % d['ms_per_step'])"
  done
done
