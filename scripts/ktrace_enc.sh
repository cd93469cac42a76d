#!/bin/bash
# rocprofv3 kernel-trace statistics of the encoder-forward probe (north_star figure) -> gpurun_out/<tag>_kstats.md
TAG=${1:-enc}; shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o kt -- python $R/bench.py --phase encoder-forward --encoder resnest50 --steps 10 "$@" > $R/gpurun_out/${TAG}_bench.json 2>/dev/null
cd $R
python scripts/rocpd_stats.py $(find /tmp/kt_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kstats.md > /dev/null
tail -1 gpurun_out/${TAG}_kstats.md
