#!/usr/bin/env python3
"""Is the synthesis kernel's time set by the card's power management?  The default workload's synthesis alone, back to
back (duty 100 %) and with the host sleeping `pause` microseconds between launches (the GPU idle meanwhile): if the
kernel runs faster after a pause, what limits it in a running sweep is not the kernel.
    python tools/duty_probe.py 0 100 300 1000"""
import sys
import time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath

wl = 580e-9
lens, x, u = bench.build_workload(4096, 512, 1e-3, 0.5, wl, 1.0)
src = (0.0, 0.0, -lens['source_distance'], 'x')
ctx = _lib.default_context()
hp = HotPath(src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx)
for _ in range(200):
    hp.step()
hp.sync()
for pause in [int(a) for a in sys.argv[1:]] or [0, 100, 300, 1000]:
    for what in ('nearfield', 'transform'):
        fn = hp.queue_synthesis if what == 'nearfield' else hp.queue_transform
        for _ in range(300):
            fn()
        hp.sync()
        ctx.profile(True, kernels=None, every=1)
        ctx.profile_reset()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 1.5:
            fn()
            if pause:
                hp.sync()
                t1 = time.perf_counter()
                while time.perf_counter() - t1 < pause * 1e-6:
                    pass
            n += 1
        hp.sync()
        prof = ctx.profile_get()
        ctx.profile(False)
        print('pause %5d us  %-9s loop  launches %6d  kernels ms %s' % (
            pause, what, n, {k: round(v['total_ms'] / v['launches'], 4) for k, v in prof.items() if v['launches']}), flush=True)
