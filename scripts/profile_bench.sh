#!/bin/bash
# Run ON the GPU box from the repo root: kernel-trace statistics and the two HBM-traffic PMC passes of the bench
# command (separate runs, as the MI355X guide prescribes).  Outputs land in gpurun_out/prof_<tag>/.
# usage: scripts/profile_bench.sh <tag> [extra bench args]
set -e
TAG=${1:-r01}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encoder-probe --no-split-check "$@" > $OUT/bench_under_trace.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encoder-probe --no-prof --no-split-check "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encoder-probe --no-prof --no-split-check "$@" > /dev/null 2>&1
cd $R
python scripts/rocpd_stats.py /tmp/kt/kt_results.db $OUT/kernel_stats.md > /dev/null
python scripts/pmc_traffic.py /tmp/pf/pf_results.db /tmp/pw/pw_results.db $OUT/pmc_traffic.json > /dev/null
ls /tmp/kt | head; ls -la $OUT
