#!/usr/bin/env python3
"""Every rank's shard of an N-GPU STRONG-scaling run, one after the other on a single GPU (no
all-reduce): per-rank kernel times -> the speed-up the decomposition allows before communication,
T(1 GPU) / max over ranks.  Default workload = BASELINE configs[2] (8192^2 -> 512^2, 2 mm NA 0.94).

    python tools/shard_probe.py WORLD [--sharding auto|mirrored] [--aperture N --farfield M --diameter D --na NA]
                                [--out profiles/r03_shard8.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from metalens_amd import _lib  # noqa: E402
from metalens_amd.pipeline import HotPath  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('world', type=int)
ap.add_argument('--sharding', default='auto,mirrored')
ap.add_argument('--aperture', type=int, default=8192)
ap.add_argument('--farfield', type=int, default=512)
ap.add_argument('--diameter', type=float, default=2e-3)
ap.add_argument('--na', type=float, default=0.94)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--out', default=None)
args = ap.parse_args()

t0 = time.time()
lens, x, u = bench.build_workload(args.aperture, args.farfield, args.diameter, args.na, 580e-9, 1.0)
print('aperture %d -> %d, lens %.2f mm NA %.2f: %d rings, %d cells, setup %.1f s' % (
    args.aperture, args.farfield, args.diameter * 1e3, args.na,
    len(lens['lens_periphery_summary']['r_center_list']), len(lens['lens_center_summary']), time.time() - t0))
source = (0.0, 0.0, -lens['source_distance'], 'x')
ctx = _lib.Context(0)


def timed(hp):
    for _ in range(3):
        hp.step_local()
    hp.sync()
    _lib.check(ctx.lib.ml_farfield_project_async(ctx.handle, hp.Z0))
    ctx.profile(True)
    ctx.profile_reset()
    t = time.perf_counter()
    for _ in range(args.steps):
        hp.step_local()
        _lib.check(ctx.lib.ml_farfield_project_async(ctx.handle, hp.Z0))   # the local half of project_reduce
    hp.sync()
    dt = (time.perf_counter() - t) / args.steps
    p = ctx.profile_get()
    ctx.profile(False)
    return 1e3 * dt, {k: v['total_ms'] / args.steps for k, v in p.items() if v['launches']}


hp = HotPath(source, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'],
             x, x, u, u, ctx=ctx)
whole_ms, whole_k = timed(hp)
print('1 GPU: %.3f ms/step (host clock, profiled)  %s' % (whole_ms, {k: round(v, 3) for k, v in whole_k.items()}))
report = {'workload': '%dx%d -> %dx%d, %.3g mm NA %.2g' % (args.aperture, args.aperture, args.farfield,
                                                          args.farfield, args.diameter * 1e3, args.na),
          'world': args.world, 'one_gpu': {'ms_per_step': whole_ms, 'kernels_ms': whole_k,
                                           'kernel_sum_ms': sum(whole_k.values())}, 'shardings': {}}
for sharding in args.sharding.split(','):
    ranks = []
    for rank in range(args.world):
        hp = HotPath(source, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'],
                     lens['hexgridset'], x, x, u, u, ctx=ctx, rank=rank, world=args.world, sharding=sharding)
        ms, k = timed(hp)
        ranks.append({'rank': rank, 'rows': int(hp.x_local.size), 'sharding': hp.sharding, 'ms_per_step': ms,
                      'kernels_ms': k, 'kernel_sum_ms': sum(k.values())})
        print('  %-9s rank %d: rows %d (%s)  %.3f ms/step  %s' % (
            sharding, rank, hp.x_local.size, hp.sharding, ms, {kk: round(v, 3) for kk, v in k.items()}))
    worst = max(r['kernel_sum_ms'] for r in ranks)
    report['shardings'][sharding] = {
        'ranks': ranks, 'max_rank_kernel_sum_ms': worst,
        'speedup_before_communication': report['one_gpu']['kernel_sum_ms'] / worst,
        'speedup_before_communication_host_clock': whole_ms / max(r['ms_per_step'] for r in ranks)}
    print('  %s: T1 / max_r T_r = %.2f x (kernel sums), %.2f x (host clock)' % (
        sharding, report['shardings'][sharding]['speedup_before_communication'],
        report['shardings'][sharding]['speedup_before_communication_host_clock']))
report['note'] = ('every rank\'s shard run ALONE on one GPU: near field + both transform stages + the local '
                  'projection, HIP-event kernel times; the all-reduce of the two projected amplitudes (8 MB at '
                  '512^2) is NOT in these numbers - no multi-GPU box was available to measure it')
if args.out:
    with open(args.out, 'w') as f:
        json.dump(report, f, indent=1)
