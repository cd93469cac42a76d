#!/bin/bash
# Run ON the GPU box from the repo root: the crowding stress behind DESIGN.md section 7 "race hygiene" - SIX test processes share
# the one GPU (pytest-xdist, as the suite itself runs) and repeat the comparisons that are sensitive to an inter-block hand-off
# going wrong under oversubscription: whole training steps against the oracle, bit-reproducibility, Siamese batched-vs-sequential,
# the small-grid / two-plane kernels, the statistics reductions.  A located race shows as a failure in SOME repetition; the log
# lists every repetition's counts.     usage: scripts/crowd_stress.sh [repetitions] [tag]
REPS=${1:-6}; TAG=${2:-crowd}
OUT=gpurun_out/${TAG}_stress.txt
echo "crowding stress: $REPS repetitions x 6 xdist workers on one GPU ($(date -u +%FT%TZ), commit $(git rev-parse --short HEAD 2>/dev/null))" > $OUT
echo "environment: XV2_BN_FOLD=${XV2_BN_FOLD:-unset(off)} XV2_SG=${XV2_SG:-unset(on)} XV2_F16X2=${XV2_F16X2:-unset(on)}" >> $OUT
FAILS=0
for r in $(seq 1 $REPS); do
  python -m pytest tests/test_model_gpu.py tests/test_sg_conv_gpu.py tests/test_f16x2_gpu.py tests/test_coop_gpu.py -m gpu -q -n 6 \
    -k "not fullsize" -p no:cacheprovider > /tmp/crowd_$r.log 2>&1
  LINE=$(tail -1 /tmp/crowd_$r.log)
  echo "repetition $r: $LINE" >> $OUT
  grep "^FAILED" /tmp/crowd_$r.log >> $OUT
  grep -q "failed" <<< "$LINE" && FAILS=$((FAILS + 1))
done
echo "repetitions with failures: $FAILS of $REPS" >> $OUT
cat $OUT
