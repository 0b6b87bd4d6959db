#!/bin/bash
# Same-box A/B of library builds: every variant (a .so under abl_tmp/, or "product" = the in-tree library) runs
# `bench.py --profile all` alternately, ROUNDS times, in ONE gpurun call; one compact line per run.
#     tools/ab_run.sh OUTFILE ROUNDS "<bench args>" variant1 variant2 ...
# e.g. tools/ab_run.sh gpurun_out/ab1.txt 2 "" product abl_tmp/lib_x.so
OUT=$1; ROUNDS=$2; ARGS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $(dirname $OUT)
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    # variant = LIB[:NAME=VALUE[,NAME=VALUE...]] - environment for a -DML_DIAG build's knobs (common.h diag_int)
    lib=${v%%:*}; envs=""; [ "$lib" != "$v" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    if [ "$lib" = product ]; then unset METALENS_HIP_LIB; else export METALENS_HIP_LIB=$R/$lib; fi
    env $envs timeout 300 python $R/bench.py --profile all --steps 20 --blocks 4 --cpu-rows 0 --cpu-fft-side 0 --cold 0 --also-physical 0 $ARGS 2>>$OUT.err | python -c "
import json,sys
v=sys.argv[1]
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    k={a:round(b,4) for a,b in d.get('kernels_ms_per_step',{}).items()}
    e=d.get('rel_err') or {}
    print('%-28s blocks %s kernels %s nf_err %.2e ff_pw %.2e' % (v, [round(b,4) for b in d['ms_per_step_blocks']], k, e.get('nearfield_vs_oracle',-1), e.get('farfield_E_pointwise_above_1e-3_of_peak',-1)))
" "$v" >> $OUT
  done
done
cat $OUT
