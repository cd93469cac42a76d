#!/bin/bash
# Run ON the GPU box: ONE test, repeated by six concurrent processes (maximum crowding of the GPU by identical work).
# usage: scripts/crowd_one.sh <pytest node id or -k expr file> <reps per process> [tag]      environment passes through
T=$1; REPS=${2:-3}; TAG=${3:-one}
OUT=gpurun_out/${TAG}_crowd_one.txt
echo "six processes x $REPS repetitions of $T; env: $(env | grep '^XV2_' | tr '\n' ' ')" > $OUT
for w in 1 2 3 4 5 6; do
  ( for r in $(seq 1 $REPS); do
      python -m pytest "$T" -m gpu -q -p no:xdist -p no:cacheprovider > /tmp/co_${w}_$r.log 2>&1
      echo "worker $w rep $r: $(tail -1 /tmp/co_${w}_$r.log) $(grep '^FAILED' /tmp/co_${w}_$r.log | sed 's/.*\[//' | tr '\n' ' ')"
    done ) >> $OUT &
done
wait
echo "failures: $(grep -c failed $OUT) of $((6 * REPS))" >> $OUT
tail -3 $OUT
