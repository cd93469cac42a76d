#!/bin/bash
# clock / pipe-duty counters of the dec2.c1 forward launch for the full and ablated libraries (run ON the GPU box):
#   scripts/pmc_abl.sh libxv2.so abl/xv2_1.so abl/xv2_3.so ...
R=$PWD
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  echo "#### $L"
  rm -rf /tmp/pmcA /tmp/pmcB
  XV2_LIB=$R/xview2_amd/$L rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmcA -o a -- python $R/scripts/one_conv.py dec2 fwd 10 >/dev/null 2>&1
  XV2_LIB=$R/xview2_amd/$L rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmcB -o b -- python $R/scripts/one_conv.py dec2 fwd 10 >/dev/null 2>&1
  (cd $R && python scripts/rocpd_pmc.py /tmp/pmcA/a_results.db igemm_kernel && python scripts/rocpd_pmc.py /tmp/pmcB/b_results.db igemm_kernel)
done
