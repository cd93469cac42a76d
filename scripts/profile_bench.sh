#!/bin/bash
# Run ON the GPU box from the repo root: kernel-trace statistics and the two HBM-traffic PMC passes of the bench
# command (separate runs, as the MI355X guide prescribes).  Outputs land in gpurun_out/prof_<tag>/.
# usage: XV2_COMMIT=<short hash> scripts/profile_bench.sh <tag> [extra bench args]
TAG=${1:-r01}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoder-probe --no-split-check --no-other-configs "$@" > $OUT/bench_under_trace.json 2>/dev/null
(cd $R && python scripts/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $OUT/kernel_stats.md > /dev/null)
# the PMC passes (the headline configuration only; a pass that dies leaves the traffic file unwritten, nothing else)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encoder-probe --no-prof --no-split-check --no-other-configs "$@" > /dev/null 2>&1 || echo "FETCH_SIZE pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encoder-probe --no-prof --no-split-check --no-other-configs "$@" > /dev/null 2>&1 || echo "WRITE_SIZE pass failed"
cd $R
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python scripts/pmc_traffic.py $F $W $OUT/pmc_traffic.json > /dev/null; fi
ls -la $OUT
