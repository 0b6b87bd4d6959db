#!/bin/bash
# second SQ counter set (instruction mix, LDS): rocprofv3 PMC pass (own run, kernel-trace only).
# usage: tools/pmc_pass2.sh OUTDIR -- <command>
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out -- "$@"
