#!/bin/bash
# usage (ON the GPU box): scripts/ab_env.sh VAR v1 v2 ... -- [bench args]   -> img/s and ms/step of bench.py per value of VAR
VAR=$1; shift
VALS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VALS+=("$1"); shift; done
shift
for v in "${VALS[@]}"; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-encoder-probe --no-prof --steps 20 --warmup 5 "$@" 2>/dev/null | VAL=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['VAL'], d['value'], d['ms_per_step'])"
done
