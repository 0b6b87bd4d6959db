#!/usr/bin/env python3
"""Digest the rocprofv3 counter passes of ONE bench.py configuration into an entry of
profiles/pmc_table.json, which bench.py reads for `roofline.traffic`, `roofline.valu` etc.

    python tools/pmc_table.py KEY FETCH_DIR WRITE_DIR SQ_DIR [--table profiles/pmc_table.json]

KEY is bench.py's config key, e.g. "gpus=1,aperture=4096,farfield=512,precision=f64,method=auto,zoom=1,pols=1"
(bench.py prints it as `config.pmc_key`).  The three directories hold the output of

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d FETCH_DIR -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE ...            (separate passes, kernel-trace only: MI355X_MICROARCH.md)
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE ...

Per kernel class (near field, stage 1, stage 2, projection) the STEADY-STATE launch shape is taken:
the (kernel, grid) pair with the most dispatches - for the near field, whose step is TWO launches (ring
and centre kernel), the steady-state shape of each kernel, summed.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch;
FETCH_SIZE is doubled (gfx950 counts a 128-byte request of a wide coalesced read as 64 bytes)."""
import collections
import csv
import glob
import json
import os
import sys

CLASSES = (('nearfield', ('nearfield_ring_kernel', 'nearfield_centre_kernel', 'nearfield_field_kernel')),
           ('stage1', ('zfft_kernel<16, 256, 2, 1', 'zfft_kernel<8, 128, 2, 1', 'zfft_kernel<32, 512, 2, 1',
                       'zfft_kernel<4, 64, 2, 1', 'zfft_kernel<0, 512, 1, 1', 'zfft_multi_kernel<1>',
                       'zfft_pass_kernel<16, 2, 2, 2, 1>', 'zfft_pass_kernel<32, 2, 2, 2, 1>')),
           ('stage2', ('zfft_kernel<16, 256, 2, 2', 'zfft_kernel<8, 128, 2, 2', 'zfft_kernel<32, 512, 2, 2',
                       'zfft_kernel<4, 64, 2, 2', 'zfft_kernel<0, 512, 1, 2', 'zfft_multi_kernel<2>',
                       'zfft_interleaved_kernel', 'zfft_pass_kernel<16, 2, 2, 2, 2>',
                       'zfft_pass_kernel<32, 2, 2, 2, 2>',
                       # (a transposed stage-1 result: stage 2 reads contiguous rows, PASS = 3)
                       'zfft_kernel<16, 256, 2, 3', 'zfft_kernel<8, 128, 2, 3', 'zfft_kernel<32, 512, 2, 3',
                       'zfft_pass_kernel<16, 2, 2, 2, 3>', 'zfft_pass_kernel<32, 2, 2, 2, 3>')),
           ('project', ('project_kernel',)))


def rows(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def per_class(d):
    """{class: {counter: average per dispatch of the most frequent (kernel, grid)}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(d):
        name = r['Kernel_Name'].replace('void ', '')
        for cls, pats in CLASSES:
            if any(('ml::' + p) in name for p in pats):
                acc[(cls, name.split('(')[0], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {}
    for cls, _ in CLASSES:
        # one steady-state shape per kernel INSTANTIATION of the class (the two ring-kernel instantiations of a
        # lens with narrow and wide collections both run every step), summed; instantiations that ran only a
        # few times (the full-grid first pass on a geometry) are not steady state
        by_kernel = collections.defaultdict(list)
        for k, c in acc.items():
            if k[0] == cls:
                by_kernel[k[1]].append((max(len(v) for v in c.values()), k))
        if not by_kernel:
            continue
        most = max(max(cands)[0] for cands in by_kernel.values())
        by_kernel = {n: cands for n, cands in by_kernel.items() if 2 * max(cands)[0] >= most}
        total, names, grid = collections.defaultdict(float), [], 0
        for cands in by_kernel.values():
            _, best = max(cands)
            for n, v in acc[best].items():
                total[n] += sum(v) / len(v)
            names.append(best[1])
            grid += int(best[2])
        out[cls] = dict(total)
        out[cls]['_kernel'] = ' + '.join(sorted(names))
        out[cls]['_grid'] = grid
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    table = 'profiles/pmc_table.json'
    if '--table' in sys.argv:
        table = sys.argv[sys.argv.index('--table') + 1]
        args = [a for a in args if a != table]
    key, fetch_dir, write_dir, sq_dir = args[:4]
    fetch, write, sq = per_class(fetch_dir), per_class(write_dir), per_class(sq_dir)
    entry = {}
    for cls, _ in CLASSES:
        e = {}
        if cls in fetch and cls in write:
            e['fetch_bytes'] = 2 * 1024 * fetch[cls]['FETCH_SIZE']
            e['write_bytes'] = 1024 * write[cls]['WRITE_SIZE']
            e['traffic_bytes'] = e['fetch_bytes'] + e['write_bytes']
            e['kernel'] = fetch[cls]['_kernel']
        if cls in sq:
            for n in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE',
                      'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_LDS'):
                if n in sq[cls]:
                    e[n] = sq[cls][n]
        if e:
            entry[cls] = e
    if all('traffic_bytes' in entry.get(c, {}) for c in ('nearfield', 'stage1', 'stage2', 'project')):
        entry['step_traffic_bytes'] = sum(entry[c]['traffic_bytes'] for c in ('nearfield', 'stage1', 'stage2', 'project'))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    entry['source_id'] = bench.kernel_source_id()   # the kernels these counters belong to (bench.py flags a mismatch)
    entry['source'] = 'rocprofv3 --pmc passes digested by tools/pmc_table.py from %s, %s, %s' % (
        fetch_dir, write_dir, sq_dir)
    data = json.load(open(table)) if os.path.exists(table) else {}
    data[key] = entry
    json.dump(data, open(table, 'w'), indent=1, sort_keys=True)
    print(json.dumps({key: entry}, indent=1))


if __name__ == '__main__':
    main()
