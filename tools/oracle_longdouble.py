#!/usr/bin/env python3
"""How exact is the fp64 oracle itself?  (the figure behind the 2e-12 pointwise
bound of tests/test_gpu_parity.py::test_north_star_size_properties.)

The far-field oracle (oracle/farfield_oracle.py) sums N^2 aperture samples per direction in fp64
(BLAS zgemm, twiddles reduced in long double and rounded to fp64).  Here the same sums are taken
once more in x87 long double (64-bit mantissa, twiddles kept in long double) on the bench workload
scaled to `--side`, and the two are compared the way the GPU is compared with the oracle:
   max |d| / max |E|                 (the suite's TOL normalisation)
   max |d| / |E| over |E| > 1e-3 max (pointwise, dim directions included)
CPU only; no GPU library is touched.  python tools/oracle_longdouble.py --side 1024

--gpu-dump FILE: the far field a GPU run of the SAME workload ended with (`bench.py --aperture SIDE
--farfield M --diameter D --na NA --dump FILE` on the GPU box; a_theta / a_phi on the whole M x M grid) is
compared with the long-double sums too: where the GPU is, not only where the oracle is.
The long-double sums start from the ORACLE's fp64 near field, so the GPU's own near-field rounding (1.4e-15 of
max|F| against the oracle) is part of what is measured."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--side', type=int, default=1024)
    ap.add_argument('--farfield', type=int, default=128)
    ap.add_argument('--diameter', type=float, default=2.5e-4)
    ap.add_argument('--na', type=float, default=0.5)
    ap.add_argument('--dirs', type=int, default=16)
    ap.add_argument('--gpu-dump', default=None)
    ap.add_argument('--orders', default='survey', help="bench.py --orders of the workload ('survey' | 'physical')")
    args = ap.parse_args()
    import bench
    from oracle import farfield_oracle as fo, nearfield_oracle as no
    wl = 580e-9
    lens, x, u = bench.build_workload(args.side, args.farfield, args.diameter, args.na, wl, 1.0, orders=args.orders)
    t0 = time.time()
    F = no.build_nearfield(0.0, 0.0, -lens['source_distance'], 'x', wl, lens['lens_periphery_summary'],
                           lens['lens_center_summary'], lens['hexgridset'], x_pts=x, y_pts=x)
    n_glass = F[7]
    F = F[:4]
    t1 = time.time()
    sel = np.arange(0, u.size, max(1, u.size // args.dirs))
    us = u[sel]
    ref64 = fo.farfield_direct(*F, x, x, wl, n_glass, us, us)

    def tw(n, step, uu):   # axis_twiddles without the final rounding to fp64
        uu = np.asarray(uu, dtype=np.longdouble).reshape(-1, 1)
        pos = (np.arange(n) - (n - n // 2)).astype(np.longdouble).reshape(1, -1) * np.longdouble(step)
        turns = pos * uu * (np.longdouble(n_glass) / np.longdouble(wl))
        turns = turns - np.rint(turns)
        ang = (-2 * np.pi * np.longdouble(1)) * turns
        # np.pi is fp64: use the long double constant
        ang = turns * (-2 * np.longdouble('3.14159265358979323846264338327950288'))
        return np.cos(ang) + 1j * np.sin(ang)

    dx = np.longdouble(x[1]) - np.longdouble(x[0])
    A, B = tw(x.size, dx, us), tw(x.size, dx, us)
    dA = dx * dx

    def transform(Fk):
        return (A @ Fk.astype(np.clongdouble)) @ B.T

    Nx, Ny, Lx, Ly = (-transform(F[3]) * dA, transform(F[2]) * dA, transform(F[1]) * dA, -transform(F[0]) * dA)
    _, ath, aph = fo.project(Nx, Ny, Lx, Ly, us, us, wl, n_glass, return_amplitudes=True)
    t2 = time.time()
    out = {'side': args.side, 'orders': args.orders, 'directions': int(sel.size) ** 2, 'nearfield_s': t1 - t0, 'sums_s': t2 - t1}
    for key, ld in (('a_theta', ath), ('a_phi', aph)):
        d = np.abs(ref64[key].astype(np.clongdouble) - ld)
        mag = np.abs(ld)
        bright = mag > 1e-3 * mag.max()
        out[key] = {'rel_to_max': float(d.max() / mag.max()),
                    'pointwise_above_1e-3_of_peak': float((d[bright] / mag[bright]).max()),
                    'bright_directions': int(bright.sum())}
    if args.gpu_dump:
        z = np.load(args.gpu_dump)
        assert z['a_theta'].shape == (u.size, u.size), 'the dump belongs to another direction grid'
        out['gpu_dump'] = os.path.basename(args.gpu_dump)
        for key, ld in (('a_theta', ath), ('a_phi', aph)):
            g = z[key][np.ix_(sel, sel)].astype(np.clongdouble)
            d, d64 = np.abs(g - ld), np.abs(g - ref64[key].astype(np.clongdouble))
            mag = np.abs(ld)
            bright = mag > 1e-3 * mag.max()
            out[key]['gpu_vs_longdouble'] = {'rel_to_max': float(d.max() / mag.max()),
                                             'pointwise_above_1e-3_of_peak': float((d[bright] / mag[bright]).max())}
            out[key]['gpu_vs_oracle'] = {'rel_to_max': float(d64.max() / mag.max()),
                                         'pointwise_above_1e-3_of_peak': float((d64[bright] / mag[bright]).max())}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
