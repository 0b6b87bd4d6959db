#!/bin/bash
# One round's evidence in one GPU call: rocprofv3 kernel trace + stats, the HBM-traffic and SQ
# counter passes (each in its own run, kernel-trace only, as MI355X_MICROARCH.md prescribes) and
# the bench lines at the configurations DESIGN.md quotes.  Run on the GPU box:
#     tools/profile_round.sh TAG        -> gpurun_out/TAG/...
# EVERY rocprofv3 call runs under `timeout`: a counter set the profiler cannot collect makes it
# abort and then wait forever (round 2 lost 40 GPU-minutes to exactly that with TA_* counters).
# then digest locally with tools/profile_summary.py / tools/pmc_digest.py into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
CMD="python $R/bench.py --steps 7 --warmup 2 --cpu-rows 0"
cd /tmp && export TMPDIR=/tmp
# the kernel-trace pass runs EXACTLY the default command (what the driver runs at N = 1)
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py > $O/stats.log 2>&1
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
bash $R/tools/pmc_mfma.sh $O/mfma -- $CMD > $O/mfma.log 2>&1
bash $R/tools/pmc_pass2.sh $O/sq2 -- $CMD > $O/sq2.log 2>&1
cd $R
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python bench.py --profile all --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_profile_all.json
python bench.py --profile none --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_profile_none.json
python bench.py --aperture 4096 --farfield 512 --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_4096x512.json
python bench.py --aperture 4096 --farfield 512 --cpu-rows 0 --profile all 2>/dev/null | tail -1 > $O/bench_4096x512_profile_all.json
python bench.py --precision f32 --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_f32.json
python bench.py --precision f32 --aperture 4096 --farfield 512 --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_f32_4096x512.json
python bench.py --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94 --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_8192x512_na094.json
ML_NO_PLAN_CACHE=1 ML_EAGER_UNFOLD=1 python bench.py --profile none --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_no_plan_cache.json
python bench.py --warmup 0 --steps 2 --cpu-rows 0 2>/dev/null | tail -1 > $O/bench_warmup0.json
ls -la $O
