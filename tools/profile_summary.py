#!/usr/bin/env python3
"""Digest rocprofv3 CSV output (one --kernel-trace --stats run, optional --pmc FETCH_SIZE and
--pmc WRITE_SIZE runs of the SAME command) into one small text summary for profiles/.

    python tools/profile_summary.py STATS_DIR [FETCH_DIR] [WRITE_DIR] [SQ_DIR] [LABEL=SQ_DIR ...] > profiles/rNN_summary.txt

Counters are per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB (checked here: the near-field
kernel's WRITE_SIZE equals its 64 B/sample of stores).  Per MI355X_MICROARCH.md §HBM, on gfx950
FETCH_SIZE counts 128-byte requests at 64 bytes for wide (16 B/lane) coalesced reads, so the
"fetch x2" column is the corrected figure for kernels that read that way (zgemm A/B tiles).
"""
import collections
import csv
import glob
import os
import sys


def find(d, pat):
    hits = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace('void ', '')
    return name.split('(')[0][:44]


def main():
    stats_dir = sys.argv[1]
    out = []
    ks = find(stats_dir, '*kernel_stats.csv')
    out.append('== rocprofv3 --kernel-trace --stats : %s' % os.path.basename(ks))
    out.append('%-46s %6s %12s %12s %7s' % ('kernel', 'calls', 'avg_us', 'total_us', '%'))
    for r in csv.DictReader(open(ks)):
        out.append('%-46s %6s %12.1f %12.1f %7s' % (short(r['Name']), r['Calls'],
                                                    float(r['AverageNs']) / 1e3,
                                                    float(r['TotalDurationNs']) / 1e3, r['Percentage']))
    # per (kernel, grid) durations: separates the two zgemm stages
    kt = find(stats_dir, '*kernel_trace.csv')
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        groups[(short(r['Kernel_Name']), r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'])].append(
            (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out.append('')
    out.append('== per (kernel, grid) dispatch durations')
    out.append('%-46s %14s %6s %12s %12s' % ('kernel', 'grid', 'calls', 'avg_us', 'min_us'))
    for (k, g), v in sorted(groups.items()):
        out.append('%-46s %14s %6d %12.1f %12.1f' % (k, g, len(v), sum(v) / len(v), min(v)))
    for label, d in zip(('FETCH_SIZE', 'WRITE_SIZE'), sys.argv[2:4]):
        cc = find(d, '*counter_collection.csv')
        # grid shape per dispatch from the same run's kernel trace (separates launches of one
        # kernel that have the same total size, e.g. the two folded GEMM stages)
        shape = {}
        kt2 = find(d, '*kernel_trace.csv')
        if kt2:
            for r in csv.DictReader(open(kt2)):
                shape[r['Dispatch_Id']] = r['Grid_Size_X'] + 'x' + r['Grid_Size_Y']
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(cc)):
            acc[(short(r['Kernel_Name']), shape.get(r['Dispatch_Id'], r['Grid_Size']))].append(
                float(r['Counter_Value']))
        out.append('')
        out.append('== rocprofv3 --pmc %s (KiB per dispatch)' % label)
        out.append('%-46s %14s %6s %14s %14s' % ('kernel', 'grid', 'calls', 'avg_KiB', 'avg_MB(x2)' if label == 'FETCH_SIZE' else 'avg_MB'))
        for (k, g), v in sorted(acc.items()):
            avg = sum(v) / len(v)
            mb = avg * 1024 / 1e6 * (2 if label == 'FETCH_SIZE' else 1)
            out.append('%-46s %14s %6d %14.1f %14.2f' % (k, g, len(v), avg, mb))
    # further directories: the SQ_* pass of the same command, then LABEL=DIR pairs of other configurations
    for arg in sys.argv[4:]:
        label, d = arg.split('=', 1) if '=' in arg else ('the same command', arg)
        cc = find(d, '*counter_collection.csv')
        if not cc:
            continue
        dur = collections.defaultdict(list)
        kt2 = find(d, '*kernel_trace.csv')
        if kt2:
            for r in csv.DictReader(open(kt2)):
                dur[(short(r['Kernel_Name']), r['Grid_Size'] if 'Grid_Size' in r else r['Grid_Size_X'])].append(
                    (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(cc)):
            acc[(short(r['Kernel_Name']), r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
        out.append('')
        out.append('== rocprofv3 --pmc SQ_* of %s (per dispatch; quad-cycle units for SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_*)' % label)
        for (k, g), counters in sorted(acc.items()):
            if not k.startswith('ml::'):
                continue
            out.append('%-46s grid %s' % (k, g))
            for n, v in sorted(counters.items()):
                out.append('    %-34s %14.0f' % (n, sum(v) / len(v)))
    print('\n'.join(out))


if __name__ == '__main__':
    main()
