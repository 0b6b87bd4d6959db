#!/bin/bash
# One round's evidence in one GPU call: rocprofv3 kernel trace + stats of EXACTLY the default
# bench command (what the driver runs at N = 1), the HBM-traffic counter passes (each in its own
# run, kernel-trace only, as MI355X_MICROARCH.md prescribes) and the bench lines at the
# configurations DESIGN.md quotes.  Run on the GPU box:
#     tools/profile_round.sh TAG        -> gpurun_out/TAG/...
# then digest locally with tools/profile_summary.py / tools/pmc_digest.py into profiles/.
# EVERY rocprofv3 call runs under `timeout`: a counter set the profiler cannot collect makes it
# abort and then wait forever (round 2 lost 40 GPU-minutes to exactly that with TA_* counters).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
T="timeout -k 5 ${PMC_TIMEOUT:-240}"
SHORT="python $R/bench.py --steps 7 --warmup 2 --blocks 1 --cpu-rows 0 --cpu-fft-side 0 --check 0"
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --cpu-rows 0 --cpu-fft-side 0 > $O/stats.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $SHORT > $O/fetch.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $SHORT > $O/write.log 2>&1
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $SHORT > $O/sq.log 2>&1
$T rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq2 -- $SHORT > $O/sq2.log 2>&1
cd $R
B="python bench.py --cpu-rows 0 --cpu-fft-side 0"
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench.json
timeout 300 $B --profile all 2>/dev/null | tail -1 > $O/bench_profile_all.json
timeout 300 $B --profile none 2>/dev/null | tail -1 > $O/bench_profile_none.json
timeout 300 $B --method gemm --profile all 2>/dev/null | tail -1 > $O/bench_gemm_profile_all.json
timeout 300 $B --aperture 2048 --farfield 256 --profile all 2>/dev/null | tail -1 > $O/bench_2048x256.json
timeout 300 $B --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94 --profile all 2>/dev/null | tail -1 > $O/bench_8192x512_na094.json
timeout 300 $B --pols xyz --profile all 2>/dev/null | tail -1 > $O/bench_pols_xyz.json
timeout 300 $B --precision f32 --profile all 2>/dev/null | tail -1 > $O/bench_f32.json
timeout 300 $B --zoom 0.5 --profile all 2>/dev/null | tail -1 > $O/bench_zoom05.json
timeout 300 $B --zoom 0.7 --profile all 2>/dev/null | tail -1 > $O/bench_zoom07.json
timeout 300 $B --pair-list 4096 --profile all 2>/dev/null | tail -1 > $O/bench_pairlist4096.json
ls -la $O
