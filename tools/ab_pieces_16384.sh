#!/bin/bash
# Stage 1 of 16384^2 -> 1024^2 with its result in physical pieces of ML_G_PIECE_KB (0: hipMalloc; -1: the product rule), diagnostic build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export METALENS_HIP_LIB=$R/abl_tmp/lib_diag.so
OUT=$R/gpurun_out/pc16384.txt
for v in ${PIECES:-0 4096 16384 65536 0 8192 32768}; do
  ML_G_PIECE_KB=$v timeout 600 python $R/bench.py --aperture 16384 --farfield 1024 --diameter 4e-3 --steps 5 --warmup 1 --blocks 2 --cold 0 --profile all --cpu-rows 0 --cpu-fft-side 0 --check 0 --also-physical 0 2>>$OUT.err | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); k=d.get('kernels_ms_per_step',{})
    print('piece_kb %6s stage1 %.4f stage2 %.4f nearfield %.4f blocks %s' % (sys.argv[1], k.get('zgemm_stage1',-1), k.get('zgemm_stage2',-1), k.get('nearfield',-1), [round(b,3) for b in d['ms_per_step_blocks']]))
" $v >> $OUT
done
cat $OUT
