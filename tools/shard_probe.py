#!/usr/bin/env python3
"""Emulate ONE rank of an N-GPU weak-scaling run on a single GPU (no all-reduce): how long does
the per-rank work take for the outermost and the innermost shard?  usage: shard_probe.py WORLD"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath

world = int(sys.argv[1])
side = int(round(2048 * math.sqrt(world) / 16)) * 16
diameter = 1e-3 * side / 2048
t0 = time.time()
lens, x, u = bench.build_workload(side, 256, diameter, 0.5, 580e-9, 1.0)
print('world %d: side %d, lens %.2f mm, %d rings, %d cells, setup %.1f s' % (
    world, side, diameter * 1e3, len(lens['lens_periphery_summary']['r_center_list']),
    len(lens['lens_center_summary']), time.time() - t0))
source = (0.0, 0.0, -lens['source_distance'], 'x')
ctx = _lib.Context(0)
for rank in sorted({0, world // 2, world - 1}):
    hp = HotPath(source, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'],
                 lens['hexgridset'], x, x, u, u, ctx=ctx, rank=rank, world=world)
    hp.world = 1          # skip the all-reduce, keep this rank's shard
    for _ in range(3):
        hp.step()
    hp.sync()
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(10):
        hp.step()
    hp.sync()
    dt = (time.perf_counter() - t0) / 10
    p = ctx.profile_get(); ctx.profile(False)
    print('  rank %d: rows %d  %.3f ms/step  %s' % (rank, hp.x_local.size, dt * 1e3,
          {k: round(v['total_ms'] / 10, 3) for k, v in p.items() if v['launches']}))
