#!/bin/bash
# Run ON the GPU box: the bit-reproducibility / batched-vs-sequential comparisons of tests/test_model_gpu.py, repeated under crowding
# (six xdist workers on the one GPU, each test duplicated so that all workers stay busy).   usage: scripts/crowd_repro.sh [reps] [tag]
REPS=${1:-6}; TAG=${2:-repro}
OUT=gpurun_out/${TAG}_crowd.txt
echo "crowded reproducibility runs: $REPS repetitions, env XV2_SG=${XV2_SG:-on} XV2_F16X2=${XV2_F16X2:-on}" > $OUT
F=0
for r in $(seq 1 $REPS); do
  python -m pytest tests/test_model_gpu.py tests/test_model_gpu.py tests/test_model_gpu.py -m gpu -q -n 6 -p no:cacheprovider \
     -k "reproducible or sequential_passes or layer_level or bit" > /tmp/cr_$r.log 2>&1
  L=$(tail -1 /tmp/cr_$r.log); echo "rep $r: $L" >> $OUT; grep "^FAILED" /tmp/cr_$r.log >> $OUT
  grep -q failed <<< "$L" && F=$((F+1))
done
echo "repetitions with failures: $F of $REPS" >> $OUT
cat $OUT
