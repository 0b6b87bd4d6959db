#!/bin/bash
# generic PMC pass: tools/pmc_pass.sh OUTDIR "COUNTER LIST" -- <command>
out=$1; ctr=$2; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- "$@"
