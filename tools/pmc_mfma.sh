#!/bin/bash
# MFMA-pipe occupancy of the GEMM kernels: rocprofv3 PMC pass (own run, kernel-trace only).
# usage: tools/pmc_mfma.sh OUTDIR -- <command>
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out -- "$@"
