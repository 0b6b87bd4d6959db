#!/usr/bin/env python3
"""How much do the synthesis and the row transform really run side by side?  Digest a
`rocprofv3 --kernel-trace` of a banded / pipelined bench run: per row-transform dispatch, the
fraction of its duration during which a synthesis dispatch was executing too.

    python tools/overlap_trace.py TRACE_DIR"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
nf, s1 = [], []
for r in csv.DictReader(open(f)):
    name, a, b = r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'nearfield_field_kernel' in name:
        nf.append((a, b))
    elif 'zfft_kernel' in name and ', 1, ' in name.split('zfft_kernel')[1][:24]:
        s1.append((a, b))
nf.sort()
s1.sort()
# steady state: skip the first fifth of the run
t_lo = min(a for a, _ in nf) + (max(b for _, b in nf) - min(a for a, _ in nf)) // 5
nf = [x for x in nf if x[0] >= t_lo]
s1 = [x for x in s1 if x[0] >= t_lo]


def overlap(iv, others):
    tot = 0
    for a, b in others:
        lo, hi = max(a, iv[0]), min(b, iv[1])
        if hi > lo:
            tot += hi - lo
    return tot


frac = [overlap(iv, nf) / max(1, iv[1] - iv[0]) for iv in s1]
dur_nf = sum(b - a for a, b in nf) / max(1, len(nf)) / 1e3
dur_s1 = sum(b - a for a, b in s1) / max(1, len(s1)) / 1e3
span = (max(b for _, b in nf + s1) - min(a for a, _ in nf + s1)) / 1e3
print('%d synthesis dispatches (avg %.1f us), %d row-transform dispatches (avg %.1f us) in %.1f us of steady state'
      % (len(nf), dur_nf, len(s1), dur_s1, span))
print('row transform: on average %.0f %% of a dispatch runs while a synthesis dispatch is executing (min %.0f %%, max %.0f %%)'
      % (100 * sum(frac) / max(1, len(frac)), 100 * min(frac), 100 * max(frac)))
busy = sum(b - a for a, b in nf) + sum(b - a for a, b in s1)
print('sum of the two kernels\' durations / wall time = %.2f (1.0 = back to back, 2.0 = fully side by side)' % (busy / 1e3 / span))
