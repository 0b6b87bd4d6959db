#!/bin/bash
# The bench lines DESIGN.md section 5 quotes, one JSON file each:  tools/bench_lines.sh OUTDIR  (on the GPU box).
# Their counter-derived fields (roofline.traffic, roofline.valu) come from profiles/pmc_table.json: run this
# AFTER tools/profile_round.sh + tools/pmc_table.py so that the table belongs to the kernels on the box.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=$1
mkdir -p $O
B="python bench.py --cpu-rows 0 --cpu-fft-side 0"
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench.json
timeout 300 $B --profile all 2>/dev/null | tail -1 > $O/bench_profile_all.json
timeout 300 $B --profile none 2>/dev/null | tail -1 > $O/bench_profile_none.json
timeout 300 $B --method gemm --profile all 2>/dev/null | tail -1 > $O/bench_gemm_profile_all.json
# (2048^2: the default kernel timing - two kernels every fourth step; events around EVERY launch, --profile all, cost
# a 0.17 ms step 15 % in its first block, which is all the "first block off its median" of round 4 was)
timeout 300 $B --aperture 2048 --farfield 256 2>/dev/null | tail -1 > $O/bench_2048x256.json
timeout 300 $B --aperture 2048 --farfield 256 --profile all 2>/dev/null | tail -1 > $O/bench_2048x256_profile_all.json
timeout 300 $B --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94 --profile all 2>/dev/null | tail -1 > $O/bench_8192x512_na094.json
timeout 300 $B --orders physical --profile all 2>/dev/null | tail -1 > $O/bench_physical.json
timeout 300 $B --aperture 2048 --farfield 256 --orders physical 2>/dev/null | tail -1 > $O/bench_2048x256_physical.json
timeout 300 $B --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94 --orders physical --profile all 2>/dev/null | tail -1 > $O/bench_8192x512_na094_physical.json
timeout 300 $B --gpus 1 --scaling strong 2>/dev/null | tail -1 > $O/bench_gpus1_strong_8192.json
timeout 300 $B --positions 3 --profile all 2>/dev/null | tail -1 > $O/bench_positions3.json
timeout 300 $B --pols xyz --profile all 2>/dev/null | tail -1 > $O/bench_pols_xyz.json
timeout 300 $B --precision f32 --profile all 2>/dev/null | tail -1 > $O/bench_f32.json
timeout 300 $B --zoom 0.5 --profile all 2>/dev/null | tail -1 > $O/bench_zoom05.json
timeout 300 $B --pair-list 4096 --profile all 2>/dev/null | tail -1 > $O/bench_pairlist4096.json
timeout 600 $B --aperture 16384 --farfield 1024 --diameter 4e-3 --steps 5 --warmup 1 --blocks 2 --cold 0 --profile all 2>/dev/null | tail -1 > $O/bench_16384x1024_f64.json
# the multi-rank path on this one GPU: plain `bench.py --gpus N` starts its own ranks (file communicator)
for n in 2 4 8; do
  ML_COMM_BACKEND=file timeout 600 python bench.py --gpus $n --cpu-rows 0 --cpu-fft-side 0 --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94 --scaling strong 2>$O/bench_gpus$n.err | tail -1 > $O/bench_gpus${n}_8192_file_comm.json
done
