#!/bin/bash
# Per-kernel times of the once-per-geometry launches (geometry kernel, list scans, the all-patch first pass) for library
# variants: rocprofv3 --kernel-trace --stats of a short bench each.    tools/geo_probe.sh OUT variant...
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = product ]; then unset METALENS_HIP_LIB; else export METALENS_HIP_LIB=$R/$v; fi
  d=/tmp/geo_$(basename $v .so); rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --steps 10 --blocks 1 --cpu-rows 0 --cpu-fft-side 0 --also-physical 0 --check 0 > $R/$OUT.log 2>&1
  echo "== $v" >> $R/$OUT
  python - $d >> $R/$OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if any(k in n for k in ('geometry','active_','row_extent','<1, false','sum_partials')):
        print('  %-60s calls %4s avg %9.1f min %9.1f us' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
cat $R/$OUT
