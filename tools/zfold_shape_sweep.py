#!/usr/bin/env python3
"""Time stage 1 for an arbitrary shard shape: zfold_shape_sweep.py NXL NY MY [reps]
(ML_ZFOLD_TILE / ML_STAGE1_SPLIT force a configuration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalens_amd import _lib
import metalens_amd as ma
nxl, ny, my = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ctx = _lib.default_context()
ctx.set_precision(os.environ.get('ML_PRECISION', 'f64'))
rng = np.random.default_rng(0)
F = [(rng.standard_normal((nxl, ny)) + 1j * rng.standard_normal((nxl, ny))) for _ in range(4)]
_lib.check(ctx.lib.ml_fields_upload(ctx.handle, nxl, ny, *[_lib.dptr(a) for a in F]))
wl, n = 580e-9, 1.459
pitch = wl / 2.2
du = (wl / n) / (pitch * ny)
u = (np.arange(my) - my // 2) * du
if os.environ.get('ML_SYMMETRIC_GRID'):
    u = (np.arange(my) - (my - 1) / 2) * du   # centre-symmetric about 0: no input modulation
t = ma.FarfieldTransform(ny, ny, pitch, pitch, wl, n, u, u, ctx=ctx)
q0 = (ny - nxl) // 2 // 2
t.transform(row0=q0, mirrored=True)
ctx.profile(True); ctx.profile_reset()
for _ in range(reps):
    t.transform(row0=q0, mirrored=True)
p = ctx.profile_get()
print('tile=%s split=%s nxl=%d ny=%d my=%d: stage1 %.3f ms stage2 %.3f ms' % (
    os.environ.get('ML_ZFOLD_TILE', 'auto'), os.environ.get('ML_STAGE1_SPLIT', 'auto'), nxl, ny, my,
    p['zgemm_stage1']['total_ms'] / reps, p['zgemm_stage2']['total_ms'] / reps))
