#!/usr/bin/env python3
"""Probe: do two independent passes of the hot path overlap on one GPU when issued on two
streams (two contexts)?  Near field is latency/L1-bound with idle matrix cores, the GEMMs are
MFMA-bound with little VALU/L1 use."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath

side, M = int(sys.argv[1]), int(sys.argv[2])
steps = 20
lens, x, u = bench.build_workload(side, M, 1e-3, 0.5, 580e-9, 1.0)
source = (0.0, 0.0, -lens['source_distance'], 'x')
ctxs = [_lib.Context(0), _lib.Context(0)]
hps = [HotPath(source, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'],
               lens['hexgridset'], x, x, u, u, ctx=c) for c in ctxs]
for hp in hps:
    hp.step(); hp.sync()
for n in (1, 2):
    for hp in hps: hp.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        hps[k % n].step()
    for hp in hps: hp.sync()
    dt = time.perf_counter() - t0
    print('%d stream(s): %.3f ms per step' % (n, 1e3 * dt / steps))
