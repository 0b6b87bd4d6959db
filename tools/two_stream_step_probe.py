#!/usr/bin/env python3
"""One pass split over two contexts / streams on one GPU (pipeline.HotPath2Stream) against the
single-stream pass: two_stream_step_probe.py [APERTURE FARFIELD]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath, HotPath2Stream

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lens, x, u = bench.build_workload(side, M, 1e-3, 0.5, 580e-9, 1.0)
src = (0.0, 0.0, -lens['source_distance'], 'x')
args = (src, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u)


def timeit(hp, steps=20):
    for _ in range(3):
        hp.step()
    hp.sync()
    t = time.perf_counter()
    for _ in range(steps):
        hp.step()
    hp.sync()
    return (time.perf_counter() - t) / steps * 1e3


one = HotPath(*args, ctx=_lib.default_context())
t1 = timeit(one)
r1 = one.results()
two = HotPath2Stream(*args, ctx=_lib.default_context())
t2 = timeit(two)
r2 = two.results()
err = max(np.nanmax(np.abs(r1[k] - r2[k])) / np.nanmax(np.abs(r1[k])) for k in ('a_theta', 'a_phi', 'P'))
print('%d^2 -> %d^2: one stream %.4f ms/step, two streams %.4f ms/step (x%.3f), max rel diff %.2e, power %.6e vs %.6e'
      % (side, M, t1, t2, t1 / t2, err, r1['power_local_rows'], r2['power_local_rows']))
