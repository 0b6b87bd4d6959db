#!/bin/bash
# The two "modes" of the row transform (0.18 vs 0.20 ms per launch in different PROCESSES of one library; the
# synthesis 1.5 % the other way): N profiled processes, counters of the memory path per kernel, next to the duration.
#     tools/mode_probe.sh TAG N        -> gpurun_out/TAG/run*/..., gpurun_out/TAG/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
N=${2:-8}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="python $R/bench.py --steps 7 --warmup 2 --blocks 1 --cpu-rows 0 --cpu-fft-side 0 --check 0 --cold 0 --also-physical 0"
for k in $(seq 1 $N); do
  timeout -k 5 ${PMC_TIMEOUT:-60} rocprofv3 --pmc ${PMC:-GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum} --kernel-trace --output-format csv -d $O/run$k -- $SHORT > $O/run$k.log 2>&1
  echo "== run $k" >> $O/summary.txt
  python $R/tools/pmc_digest.py $O/run$k 2>/dev/null | grep -A9 -E "zfft_kernel|nearfield_ring_kernel" | grep -v "^--" >> $O/summary.txt
  rm -rf $O/run$k    # (the raw csv files are large)
done
cat $O/summary.txt
