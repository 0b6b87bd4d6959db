#!/bin/bash
# Same-box A/B of the measured placement of the row transform's result (farfield.hip transform_impl): alternately
# `bench.py --placement 0` and `--placement K`, ROUNDS processes each; one compact line per process.
#     tools/ab_placement.sh OUTFILE ROUNDS [K]
OUT=$1; ROUNDS=$2; K=${3:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $(dirname $OUT)
for r in $(seq 1 $ROUNDS); do
  for p in 0 $K; do
    timeout 300 python $R/bench.py --placement $p --profile all --steps 20 --blocks 4 --cpu-rows 0 --cpu-fft-side 0 --cold 0 --also-physical 0 2>>$OUT.err | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    k={a:round(b,4) for a,b in d.get('kernels_ms_per_step',{}).items() if a in ('zgemm_stage1','zgemm_stage2','nearfield')}
    e=d.get('rel_err') or {}
    print('placement %s blocks %s kernels %s %s ff_pw %.2e' % (sys.argv[1], [round(b,4) for b in d['ms_per_step_blocks']], k, d['config']['transform']['placement'], e.get('farfield_E_pointwise_above_1e-3_of_peak',-1)))
" $p >> $OUT
  done
done
cat $OUT
