#!/usr/bin/env python3
"""When does the output-pruned FFT on a PADDED lattice (an aperture whose sample count is not a multiple of 256 runs
on the 256 / gcd(N, 256) times finer lattice, zfft.hip zfft_commensurate) still beat the folded GEMMs?  (ADVICE r5:
'auto' had no cost model.)  Times the transform alone - resident random fields, `--reps` transforms between two host
clock reads - for method auto and method gemm on square apertures of `--sides` with M = `--bins` lattice directions.

    python tools/padded_fft_sweep.py --sides 400,1000,1920,2000,3000 --bins 64,256,0      (0: M = N)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WL, N_GLASS = 580e-9, 1.459


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sides', default='400,1000,1920,2000,3000')
    ap.add_argument('--bins', default='64,256,0')
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--methods', default='auto,gemm')
    args = ap.parse_args()
    from metalens_amd import _lib
    from metalens_amd.nearfield_farfield import FarfieldTransform
    ctx = _lib.default_context()
    rng = np.random.default_rng(1)
    for n in [int(s) for s in args.sides.split(',')]:
        p = WL / 2.2
        x = (np.arange(n) - (n - 1) / 2) * p
        F = [_lib.c128(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for _ in range(4)]
        _lib.check(ctx.lib.ml_fields_upload(ctx.handle, n, n, *[_lib.dptr(a) for a in F]))
        for m in [int(s) or n for s in args.bins.split(',')]:
            if m > n:
                continue
            u = (np.arange(m) - m // 2) * ((WL / N_GLASS) / ((x[1] - x[0]) * n))
            row = {'n': n, 'm': m}
            ref = None
            for method in args.methods.split(','):
                ctx.set_method(method)
                t = FarfieldTransform(n, n, x[1] - x[0], x[1] - x[0], WL, N_GLASS, u, u, ctx=ctx)
                for _ in range(5):
                    t.transform()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    _lib.check(ctx.lib.ml_farfield_transform_async(ctx.handle, 0, 0))
                ctx.sync()
                row[method + '_ms'] = round(1e3 * (time.perf_counter() - t0) / args.reps, 4)
                row[method + '_kernels'] = ctx.plan_kernels()
                info = ctx.plan_info() if hasattr(ctx, 'plan_info') else None
                if info:
                    row[method + '_plan'] = info
                v = t.radiation_vectors()['Nx']
                if ref is None:
                    ref = v
                else:
                    row['rel_diff'] = float(np.abs(v - ref).max() / np.abs(ref).max())
            print(json.dumps(row), flush=True)
    ctx.set_method('auto')


if __name__ == '__main__':
    main()
