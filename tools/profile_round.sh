#!/bin/bash
# One round's evidence in one GPU call: rocprofv3 kernel trace + stats of EXACTLY the default
# bench command (what the driver runs at N = 1), the counter passes (each in its own run,
# kernel-trace only, as MI355X_MICROARCH.md prescribes) for the configurations DESIGN.md quotes,
# and their bench lines.  Run on the GPU box:
#     tools/profile_round.sh TAG        -> gpurun_out/TAG/...
# then digest locally:  tools/profile_summary.py (text summary) and tools/pmc_table.py (the
# per-configuration counter table bench.py reads, profiles/pmc_table.json).
# EVERY rocprofv3 call runs under `timeout`: a counter set the profiler cannot collect makes it
# abort and then wait forever (round 2 lost 40 GPU-minutes to exactly that with TA_* counters).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
T="timeout -k 5 ${PMC_TIMEOUT:-240}"
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --cpu-rows 0 --cpu-fft-side 0 > $O/stats.log 2>&1
pmc() {   # KEY=<config.pmc_key> pmc NAME <bench args>: the counter passes of one configuration
  name=$1; shift
  SHORT="python $R/bench.py --steps ${PMC_STEPS:-7} --warmup 2 --blocks 1 --cpu-rows 0 --cpu-fft-side 0 --check 0 --cold 0 --also-physical 0 $@"
  $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$name/fetch -- $SHORT > $O/$name.fetch.log 2>&1
  $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/$name/write -- $SHORT > $O/$name.write.log 2>&1
  $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/$name/sq -- $SHORT > $O/$name.sq.log 2>&1
  # the instruction mix of the synthesis kernels (per-wave figures of DESIGN 4.1)
  $T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/$name/insts -- $SHORT > $O/$name.insts.log 2>&1
  # digested ON THE BOX into the counter table bench.py reads (key = bench.py's config.pmc_key), so that the bench lines
  # below carry the counters of the kernels they ran; a copy of the table goes home in $O
  (cd $R && python tools/pmc_table.py "$KEY" $O/$name/fetch $O/$name/write $O/$name/sq > $O/$name.table.log 2>&1)
  (cd $R && python tools/pmc_digest.py $O/$name/insts > $O/$name.insts.txt 2>&1)
}
K0="gpus=1,aperture=%s,farfield=%s,precision=f64,method=auto,zoom=1,pols=1"
KEY=$(printf $K0 4096 512) pmc c4096
KEY=$(printf $K0 2048 256) pmc c2048 --aperture 2048 --farfield 256
KEY=$(printf $K0 8192 512) pmc c8192 --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94
KEY=$(printf $K0 512 64) pmc c512 --aperture 512 --farfield 64 --diameter 1.2e-4
# configs[4]'s size on one GPU (the two-pass row transform): its counters too, so that its bench line quotes its own
KEY=$(printf $K0 16384 1024) PMC_STEPS=3 pmc c16384 --aperture 16384 --farfield 1024 --diameter 4e-3
# ... and with the order lists characterize() would record (7 to 11 orders per ring collection)
KEY=$(printf $K0 4096 512),orders=physical pmc c4096phys --orders physical
KEY=$(printf $K0 8192 512),orders=physical pmc c8192phys --aperture 8192 --farfield 512 --diameter 2e-3 --na 0.94 --orders physical
cp $R/profiles/pmc_table.json $O/pmc_table.json
SHORT="python $R/bench.py --steps 7 --warmup 2 --blocks 1 --cpu-rows 0 --cpu-fft-side 0 --check 0 --cold 0 --also-physical 0"
$T rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/c4096/active -- $SHORT > $O/c4096.active.log 2>&1
cd $R
# the bench lines DESIGN.md quotes (run again after tools/pmc_table.py has refreshed the counter table, so that no line says `stale`)
bash tools/bench_lines.sh $O
timeout 300 python tools/dropin_time.py > $O/dropin.log 2>&1
timeout 200 python tools/cold_timeline.py > $O/cold_timeline.txt 2>&1
# every rank's shard of configs[2] run ALONE on this GPU (what the decomposition allows before communication)
for n in 2 4 8; do timeout 300 python tools/shard_probe.py $n --sharding auto --out $O/shard$n.json > $O/shard$n.txt 2>&1; done
# (diagnostic build with the phase stamps: make -C metalens_amd/csrc EXTRA=-DML_PHASE_TIMERS BUILD=build_pt TARGET=../../abl_tmp/lib_pt.so,
# made in the build container - it travels with the snapshot)
[ -f abl_tmp/lib_pt.so ] && METALENS_HIP_LIB=abl_tmp/lib_pt.so timeout 300 python tools/nearfield_phase_timers.py 4096 > $O/phase_timers_4096.txt 2>&1
# measured parity figures of the full-size cases (tests/test_gpu_parity.py _record) and the random sweep
ML_RECORD_PARITY=$O/parity_measured.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "north_star_size_properties" > $O/parity_pytest.log 2>&1
for seed in 5 6 7; do timeout 600 python tests/extra_random_sweep.py 120 $seed survey; timeout 600 python tests/extra_random_sweep.py 120 $seed physical; done > $O/random_sweep.txt 2>&1
ls $O
