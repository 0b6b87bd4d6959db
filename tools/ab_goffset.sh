#!/bin/bash
# How does the physical layout behind the big buffers decide the kernels' times?  Diagnostic build (common.h diag_int),
# alternating processes of the default bench; one line per process.
#     tools/ab_goffset.sh OUT ROUNDS "VARIANT ..."     VARIANT = NAME=VALUE[,NAME=VALUE...] or `default`
# knobs: ML_G_PIECE_KB (stage-1 result in physical pieces of that size; 0: hipMalloc; product: 4096), ML_G_CONTIGUOUS=1,
#        ML_G_OFFSET_KB, ML_G_SKEW (pitch skew in elements), ML_F_PIECE_KB (field planes), ML_R_PIECE_KB (geometry records)
OUT=$1; ROUNDS=$2; VARS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
export METALENS_HIP_LIB=$R/abl_tmp/lib_diag.so
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    envs=""; [ "$v" != default ] && envs=$(echo "$v" | tr ',' ' ')
    env $envs timeout 300 python $R/bench.py --profile all --steps 20 --blocks 4 --cpu-rows 0 --cpu-fft-side 0 --cold 0 --also-physical 0 --check 0 "$@" 2>>$OUT.err | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    k=d.get('kernels_ms_per_step',{})
    print('%-44s stage1 %.4f stage2 %.4f nearfield %.4f blocks %s' % (sys.argv[1], k.get('zgemm_stage1',-1), k.get('zgemm_stage2',-1), k.get('nearfield',-1), [round(b,4) for b in d['ms_per_step_blocks']]))
" "$v" >> $OUT
  done
done
cat $OUT
