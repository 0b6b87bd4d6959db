#!/bin/bash
# Does the row transform's mode follow the stage-1 result's offset inside its allocation (sub-2-MiB)?  Diagnostic build,
# no placement search; alternating processes.     tools/ab_goffset.sh OUT ROUNDS "0 4 64 1024 ..."
OUT=$1; ROUNDS=$2; OFFS=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
export METALENS_HIP_LIB=$R/abl_tmp/lib_diag.so
for r in $(seq 1 $ROUNDS); do
  for o in $OFFS; do
    ML_G_OFFSET_KB=${o%%:*} ML_G_SKEW=${o##*:} timeout 300 python $R/bench.py --placement 0 --profile all --steps 20 --blocks 2 --cpu-rows 0 --cpu-fft-side 0 --cold 0 --also-physical 0 --check 0 2>>$OUT.err | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    k=d.get('kernels_ms_per_step',{})
    print('offset_kb %6s stage1 %.4f stage2 %.4f nearfield %.4f' % (sys.argv[1], k.get('zgemm_stage1',-1), k.get('zgemm_stage2',-1), k.get('nearfield',-1)))
" $o >> $OUT
  done
done
cat $OUT
