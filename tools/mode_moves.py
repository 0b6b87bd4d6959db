#!/usr/bin/env python3
"""One context, blocks of 300 steps; a -DML_DIAG library moves ONE buffer to a fresh allocation every 300 calls
(ML_MOVE_STAGE1=300 or ML_MOVE_FIELDS=300): which buffer's placement selects the row transform's mode?"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from metalens_amd import _lib  # noqa: E402
from metalens_amd.pipeline import HotPath  # noqa: E402

wl = 580e-9
lens, x, u = bench.build_workload(4096, 512, 1e-3, 0.5, wl, 1.0)
src = (0.0, 0.0, -lens['source_distance'], 'x')
ctx = _lib.default_context()
hp = HotPath(src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx)
out = []
for block in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for _ in range(100):
        hp.step()
    hp.sync()
    ctx.profile(True, kernels=None, every=1)
    ctx.profile_reset()
    for _ in range(190):
        hp.step()
    hp.sync()
    prof = ctx.profile_get()
    ctx.profile(False)
    for _ in range(10):
        hp.step()
    hp.sync()
    out.append((round(prof['zgemm_stage1']['total_ms'] / prof['zgemm_stage1']['launches'], 4),
                round(prof['nearfield']['total_ms'] / prof['nearfield']['launches'], 4)))
print(json.dumps(out))
