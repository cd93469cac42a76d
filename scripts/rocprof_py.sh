#!/bin/bash
# Run ON the GPU box from the repo root: rocprofv3 kernel-trace statistics of `python <script> <args...>` -> gpurun_out/<tag>_kstats.md
# usage: scripts/rocprof_py.sh <tag> <script> [args]     (environment variables pass through)
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python $R/"$@" > $R/gpurun_out/${TAG}_out.txt 2>&1
cd $R
python scripts/rocpd_stats.py $(find /tmp/kt_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kstats.md > /dev/null
cat gpurun_out/${TAG}_kstats.md
