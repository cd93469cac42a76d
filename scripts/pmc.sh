#!/bin/bash
# usage: scripts/pmc.sh <layer> <fwd|dgrad|wgrad> <kernel-filter>   (run ON the GPU box from the repo root)
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcA /tmp/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d /tmp/pmcA -o a -- python $R/scripts/one_conv.py "$1" $2 >/dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d /tmp/pmcB -o b -- python $R/scripts/one_conv.py "$1" $2 >/dev/null 2>&1
cd $R
python scripts/rocpd_pmc.py /tmp/pmcA/a_results.db "$3"
python scripts/rocpd_pmc.py /tmp/pmcB/b_results.db "$3"
