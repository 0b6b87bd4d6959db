#!/usr/bin/env python3
"""per-kernel averages of every counter in a rocprofv3 counter_collection.csv"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'].replace('void ', '').split('(')[0][:40], r['Grid_Size'])
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(acc):
    if 'copyBuffer' in k[0] or 'fillBuffer' in k[0]:
        continue
    print('%-42s grid %-9s avg %.1f us' % (k[0], k[1], sum(dur[k]) / len(dur[k])))
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    for n in sorted(c):
        print('    %-28s %16.0f' % (n, c[n]))
    if c.get('SQ_INSTS_VALU_MFMA_F64'):
        print('    -> MFMA busy cycles per MFMA instruction: %.1f' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_VALU_MFMA_F64']))
    if c.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        print('    -> MFMA pipe utilisation (busy / (GUI_ACTIVE/8 x 1024 SIMDs)): %.1f %%' % (100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)))
        print('    -> effective clock: %.2f GHz' % (c['GRBM_GUI_ACTIVE'] / 8 / (sum(dur[k]) / len(dur[k])) / 1e3))
