#!/bin/bash
# Run ON the GPU box: two copies of model_hold.py side by side.    usage: run_model_hold.sh <tag> <iterations> <size> [overrides]
TAG=$1; shift
OUT=gpurun_out/${TAG}_model_hold.txt
echo "two holders: model_hold.py $*; env: $(env | grep '^XV2_' | tr '\n' ' ')" > $OUT
for w in ${HOLDERS:-B}; do
  python scripts/probes/model_hold.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids\|warn" | sed "s/^/$w: /" > /tmp/mh$w.txt &
done
python scripts/probes/model_hold.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids\|warn" | sed 's/^/A: /' >> $OUT
wait
cat /tmp/mh?.txt >> $OUT
tail -12 $OUT | cut -c1-400
