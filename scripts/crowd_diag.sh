#!/bin/bash
# Run ON the GPU box: six concurrent scripts/crowd_diag.py processes (crowding by identical work), ROUNDS times over.
# usage: scripts/crowd_diag.sh <tag> <reps> [key=value ...]         environment (XV2_*, ROUNDS) passes through
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/${TAG}_crowd_diag.txt
echo "${ROUNDS:-1} x six processes x $REPS runs of the training step ($*); env: $(env | grep '^XV2_' | tr '\n' ' ')" > $OUT
for round in $(seq 1 ${ROUNDS:-1}); do
  for w in 1 2 3 4 5 6; do
    ( python scripts/crowd_diag.py $REPS "$@" 2>&1 | grep -v "Warning\|amdgpu.ids" | sed "s/^/r$round w$w /" ) >> $OUT &
  done
  wait
done
echo "processes with a differing run: $(grep -c 'DIFFERS' $OUT)" >> $OUT
grep -c DIFFERS $OUT
