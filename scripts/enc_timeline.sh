#!/bin/bash
# ON the GPU box: scripts/enc_timeline.sh <tag> [encoder] [precision]   -> gpurun_out/<tag>_timeline.txt (environment passes through)
TAG=${1:-enc}; shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
rocprofv3 --kernel-trace -d /tmp/tl_$TAG -o tl -- python $R/scripts/enc_fwd_only.py "$@" > $R/gpurun_out/${TAG}_fwd.txt 2>&1
cd $R
python scripts/enc_timeline.py $(find /tmp/tl_$TAG -name "*.db" | head -1) > gpurun_out/${TAG}_timeline.txt
head -40 gpurun_out/${TAG}_timeline.txt; tail -2 gpurun_out/${TAG}_fwd.txt
