#!/bin/bash
# Run ON the GPU box from the repo root: kernel trace of the cfg2 bench + timeline of one step -> gpurun_out/<tag>_timeline.txt, <tag>_kstats.md
TAG=${1:-tl}; shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --no-split-check "$@" > $R/gpurun_out/${TAG}_bench.json 2>/dev/null
cd $R
DB=$(find /tmp/kt_$TAG -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/${TAG}_kstats.md > /dev/null
python scripts/timeline.py $DB 6 > gpurun_out/${TAG}_timeline.txt 2>&1
head -60 gpurun_out/${TAG}_timeline.txt
