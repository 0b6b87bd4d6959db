#!/usr/bin/env python3
"""Where does the pruned-FFT far field differ from the oracle's sums?  

Runs the bench workload at `--side` on the GPU, downloads the GPU's own near field and
 - transforms it with the ORACLE (fp64 BLAS sums, twiddles reduced in long double) on the WHOLE
   M x M direction grid,
 - compares the GPU's radiation vectors and amplitudes with it, bin by bin,
 - uploads the same fields again and transforms them with the folded GEMMs (`--method gemm`): the
   second GPU evaluation of the same sums,
and writes the error maps (magnitudes, float32) to an .npz.

    python tools/parity_isolate.py --side 2048 --farfield 256 --out gpurun_out/parity_2048.npz
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pointwise(got, ref, floor=1e-3):
    big = np.abs(ref) > floor * np.abs(ref).max()
    d = np.abs(got - ref)
    q = np.where(big, d / np.maximum(np.abs(ref), 1e-300), 0.0)
    at = np.unravel_index(np.argmax(q), q.shape)
    return float(q.max()), [int(a) for a in at], float(d.max() / np.abs(ref).max())


def analyse(z):
    out = {}
    for key in ('a_theta', 'a_phi', 'Nx', 'Ny', 'Lx', 'Ly'):
        ref = z['ref_' + key]
        for tag in ('fft', 'gemm'):
            if tag + '_' + key not in z:
                continue
            got = z[tag + '_' + key]
            pw, at, rel = pointwise(got, ref)
            out['%s %s' % (tag, key)] = dict(pointwise=pw, at=at, rel_to_max=rel)
    print(json.dumps(out, indent=1))
    # structure of the error: by row / column / residue classes of the bin numbers
    for key in ('a_phi', 'Nx', 'Ly'):
        ref, got = z['ref_' + key], z['fft_' + key]
        d = np.abs(got - ref) / np.abs(ref).max()
        print(key, 'abs error / max: median %.2e  99%% %.2e  max %.2e' % (np.median(d), np.quantile(d, 0.99), d.max()))
        M = d.shape[0]
        print('  by row    (x bins) max over columns: first 8 of sorted', np.sort(d.max(axis=1))[::-1][:8])
        print('  by column (y bins) max over rows   : first 8 of sorted', np.sort(d.max(axis=0))[::-1][:8])
        for mod in (16, M // 2):
            r = np.array([d[:, k::mod].mean() for k in range(min(mod, 16))])
            print('  mean error by column class mod %d:' % mod, np.array2string(r, precision=2))
            r = np.array([d[k::mod, :].mean() for k in range(min(mod, 16))])
            print('  mean error by row    class mod %d:' % mod, np.array2string(r, precision=2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--side', type=int, default=2048)
    ap.add_argument('--farfield', type=int, default=256)
    ap.add_argument('--diameter', type=float, default=1e-3)
    ap.add_argument('--na', type=float, default=0.5)
    ap.add_argument('--out', default='gpurun_out/parity_isolate.npz')
    args = ap.parse_args()
    import bench
    from metalens_amd import _lib
    from metalens_amd.pipeline import HotPath
    from oracle import farfield_oracle
    wl = 580e-9
    side, M = args.side, args.farfield
    lens, x, u = bench.build_workload(side, M, args.diameter, args.na, wl, 1.0)
    src = (0.0, 0.0, -lens['source_distance'], 'x')
    hp_args = (src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u)
    ctx = _lib.default_context()
    keys = ('a_theta', 'a_phi', 'Nx', 'Ny', 'Lx', 'Ly')
    save = {}
    one = HotPath(*hp_args, ctx=ctx, precision='f64', method='auto')
    one.step()
    one.sync()
    r = one.results()
    print('plan kernels', ctx.plan_kernels())
    for k in keys:
        save['fft_' + k] = np.array(r[k])
    F = [np.empty((side, side), dtype=np.complex128) for _ in range(4)]
    _lib.check(ctx.lib.ml_fields_download(ctx.handle, *[_lib.dptr(a) for a in F]))
    ref = farfield_oracle.farfield_direct(*F, x, x, wl, one.n_glass, u, u)
    for k in keys:
        save['ref_' + k] = np.array(ref[k])
    two = HotPath(*hp_args, ctx=ctx, precision='f64', method='gemm')
    two.step()
    two.sync()
    r2 = two.results()
    print('plan kernels', ctx.plan_kernels())
    for k in keys:
        save['gemm_' + k] = np.array(r2[k])
    ctx.set_method('auto')
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    analyse(save)
    # (gpurun brings back 64 MiB at most: the maps go home as float32 magnitudes)
    small = {}
    for k in keys:
        ref_k = save['ref_' + k]
        small['abs_' + k] = np.abs(ref_k).astype(np.float32)
        for tag in ('fft', 'gemm'):
            small['err_%s_%s' % (tag, k)] = (np.abs(save[tag + '_' + k] - ref_k) / np.abs(ref_k).max()).astype(np.float32)
    np.savez_compressed(args.out, u=u, **small)


if __name__ == '__main__':
    main()
