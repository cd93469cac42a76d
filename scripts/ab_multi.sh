#!/bin/bash
# usage (ON the GPU box): scripts/ab_multi.sh "ENV1=a ENV2=b" "ENV1=c" ... -- [bench args]   -> value and ms/step of bench.py per environment
SETS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do SETS+=("$1"); shift; done
shift
for rep in 1 2; do
for s in "${SETS[@]}"; do
  env $s python bench.py --no-cpu-baseline --no-encoder-probe --no-prof --no-other-configs --no-split-check "$@" 2>/dev/null | TAG="$s" python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.3f /s  %8.3f ms' % (os.environ['TAG'], d['value'], d['ms_per_step']))"
done
done
