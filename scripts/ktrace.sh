#!/bin/bash
# Run ON the GPU box from the repo root: rocprofv3 kernel-trace statistics of the bench command -> gpurun_out/<tag>_kstats.md
# usage: scripts/ktrace.sh <tag> [extra bench args]     (environment variables pass through)
TAG=${1:-kt}; shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --no-split-check "$@" > $R/gpurun_out/${TAG}_bench.json 2>/dev/null
cd $R
python scripts/rocpd_stats.py $(find /tmp/kt_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kstats.md > /dev/null
tail -1 gpurun_out/${TAG}_kstats.md
