#!/usr/bin/env python3
"""Time the two GEMM stages of the far-field transform for one problem size and one forced
tile configuration (ML_ZGEMM_TILE).  Usage: zgemm_sweep.py N M [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from metalens_amd import _lib
import metalens_amd as ma

N, M = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = _lib.default_context()
rng = np.random.default_rng(0)
F = [(rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))) for _ in range(4)]
_lib.check(ctx.lib.ml_fields_upload(ctx.handle, N, N, *[_lib.dptr(a) for a in F]))
wl, n = 580e-9, 1.459
pitch = wl / 2.2
du = (wl / n) / (pitch * N)
u = (np.arange(M) - M // 2) * du
t = ma.FarfieldTransform(N, N, pitch, pitch, wl, n, u, u, ctx=ctx)
t.transform()
from oracle import farfield_oracle
sel = np.arange(0, M, max(1, M // 8))
ref = farfield_oracle.radiation_vectors(*F, np.arange(N) * pitch, np.arange(N) * pitch, wl, n, u[sel], u[sel])
got = t.radiation_vectors()
err = max(np.abs(got[k][np.ix_(sel, sel)] - r).max() / np.abs(r).max() for k, r in zip(('Nx', 'Ny', 'Lx', 'Ly'), ref))
ctx.profile(True)
ctx.profile_reset()
for _ in range(reps):
    t.transform()
p = ctx.profile_get()
s1 = p['zgemm_stage1']['total_ms'] / reps
s2 = p['zgemm_stage2']['total_ms'] / reps
f1 = 8.0 * 4 * N * N * M
f2 = 8.0 * 4 * M * N * M
print('tile=%s N=%d M=%d stage1 %.3f ms %.1f TF | stage2 %.3f ms %.1f TF | rel err %.1e' % (
    os.environ.get('ML_ZGEMM_TILE', 'auto'), N, M, s1, f1 / s1 / 1e9, s2, f2 / s2 / 1e9, err))
