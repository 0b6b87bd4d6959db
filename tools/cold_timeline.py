"""Host timeline of single steps on a sample grid the context has not seen (bench.py `cold_step`): how long
each C-ABI call of HotPath.step() keeps the host, and when the GPU has finished.  python tools/cold_timeline.py [side]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import metalens_amd as ma
from metalens_amd import _lib

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lens, x, u = bench.build_workload(side, side // 8, 1e-3 * side / 4096, 0.5, 580e-9, 1.0)
source = (0.0, 0.0, -lens['source_distance'], 'x')
ctx = _lib.default_context()
args = (source, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'])
hp = ma.HotPath(*args, x, x, u, u, ctx=ctx)
for k in range(30):
    hp.step()
hp.sync()
pitch = x[1] - x[0]
for trial in range(4):
    shift = (0.11 + 0.13 * trial) * pitch
    hp2 = ma.HotPath(*args, x + shift, x + shift, u, u, ctx=ctx)
    ctx.sync()
    lib, h = ctx.lib, ctx.handle
    for step in range(3):
        t = [time.perf_counter()]
        _lib.check(lib.ml_nearfield_premodulate(h, 1))
        hp2._plan()
        t.append(time.perf_counter())
        _lib.check(lib.ml_nearfield_async(h, _lib.byref(hp2.params), _lib.dptr(hp2.x_local), hp2.x_local.size,
                                          _lib.dptr(hp2.y), hp2.y.size))
        t.append(time.perf_counter())
        hp2._transform()
        t.append(time.perf_counter())
        _lib.check(lib.ml_farfield_project_async(h, hp2.Z0))
        t.append(time.perf_counter())
        hp2.sync()
        t.append(time.perf_counter())
        d = [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]
        print('trial %d step %d: plan %.3f  nearfield_async %.3f  transform %.3f  project %.3f  sync %.3f  = %.3f ms'
              % (trial, step, d[0], d[1], d[2], d[3], d[4], 1e3 * (t[-1] - t[0])))
