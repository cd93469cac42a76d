#!/bin/bash
# Run ON the GPU box: eager vs --graph (whole step replayed from a captured hipGraph) for one configuration.   usage: scripts/ab_graph.sh <bench args>
for rep in 1 2; do
  for G in "" "--graph"; do
    python bench.py $G --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs --no-prof --no-encoder-probe --no-split-check "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s' % ('$G' or 'eager'), 'ms_per_step %.3f' % d['ms_per_step'], 'launch', d.get('launch'), 'loss %.5f' % d['loss'])"
  done
done
