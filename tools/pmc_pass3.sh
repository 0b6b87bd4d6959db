#!/bin/bash
# memory-side counters: usage tools/pmc_pass3.sh OUTDIR "COUNTERS" -- <command>
out=$1; ctrs=$2; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- "$@"
