#!/bin/bash
# like pmc_abl.sh but with an environment assignment instead of a library:  scripts/pmc_env.sh XV2_HALO=1 [layer] [fwd|dgrad]
R=$PWD
cd /tmp && export TMPDIR=/tmp
L=${2:-dec2}; W=${3:-fwd}; F=${4:-igemm_kernel}
rm -rf /tmp/pmcA /tmp/pmcB
env $1 XV2_LIB=${XV2_LIB:-$R/xview2_amd/libxv2.so} rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmcA -o a -- python $R/scripts/one_conv.py $L $W 10 >/dev/null 2>&1
env $1 XV2_LIB=${XV2_LIB:-$R/xview2_amd/libxv2.so} rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmcB -o b -- python $R/scripts/one_conv.py $L $W 10 >/dev/null 2>&1
cd $R && python scripts/rocpd_pmc.py /tmp/pmcA/a_results.db $F && python scripts/rocpd_pmc.py /tmp/pmcB/b_results.db $F
