#!/usr/bin/env python3
"""GPU parity sweep (tests/test_gpu_parity.py runs a short one): random windows, sources and
polarisations on the 2 mm NA 0.94 lens and on a 1 mm NA 0.5 lens against the CPU oracle,
counting discrete-decision flips (ring / sector / nearest cell).

    python tests/extra_random_sweep.py [N_CASES] [SEED] [survey|physical]

``physical``: the tables hold the order lists characterize() would record per collection and direction
(synthetic.propagating_orders: up to eleven orders per ring collection) instead of (0,0), (-1,0), (+1,0).
"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import metalens_amd as ma                      # noqa: E402
from metalens_amd import layout, synthetic     # noqa: E402
from oracle import nearfield_oracle            # noqa: E402
from test_gpu_parity import field_errors       # noqa: E402


def lens_of(radius, na, wl, orders='survey'):
    extra = {} if orders == 'survey' else {'periphery_orders': 'physical', 'center_orders': 'physical'}
    return synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet), layout.make_design,
                               radius=radius, numerical_aperture=na, wavelength=wl,
                               switch_angle=12 * math.pi / 180, num_gratings=20, num_entries=12,
                               design_kwargs={'wavelength': wl}, **extra)


def run(n_cases, seed, orders='survey'):
    """returns (worst relative field error, decision flips, exact-tie samples met)"""
    rng = np.random.default_rng(seed)
    wl = 580e-9
    lenses = [lens_of(1e-3, 0.94, wl, orders), lens_of(0.5e-3, 0.5, wl, orders)]
    worst, flips_total, ties_total = 0.0, 0, 0
    for case in range(n_cases):
        lens = lenses[case % 2]
        R = float(lens['lens_periphery_summary']['r_max_list'][-1])
        f = lens['source_distance']
        n = int(rng.integers(40, 97))
        pitch = wl / rng.uniform(2.05, 2.6)
        # window centre: anywhere in the lens, with extra weight on the diagonals and the axes
        kind = case % 4
        rad = R * math.sqrt(rng.uniform(0, 1.02))
        ang = {0: rng.uniform(0, 2 * math.pi), 1: math.pi / 4 * rng.integers(0, 8),
               2: math.pi / 2 * rng.integers(0, 4), 3: rng.uniform(0, 2 * math.pi)}[kind]
        cx, cy = rad * math.cos(ang), rad * math.sin(ang)
        if kind in (1, 2):                       # exactly symmetric grids about the origin
            cx = round(cx / pitch) * pitch
            cy = round(cy / pitch) * pitch
        x = cx + (np.arange(n) - (n - 1) / 2) * pitch
        y = cy + (np.arange(n) - (n - 1) / 2) * pitch
        pol = 'xyz'[int(rng.integers(0, 3))]
        src = (float(rng.normal(0, 2e-6)), float(rng.normal(0, 2e-6)), -f * rng.uniform(0.97, 1.03), pol)
        args = (src[0], src[1], src[2], src[3], wl, lens['lens_periphery_summary'],
                lens['lens_center_summary'], lens['hexgridset'])
        try:
            dec = {}
            want = nearfield_oracle.build_nearfield(*args, x_pts=x, y_pts=y, decisions=dec)
        except ValueError:
            try:
                ma.build_nearfield(*args, x_pts=x, y_pts=y)
            except ValueError:
                continue                          # both refuse (sample outside the tables): fine
            raise AssertionError('case %d: the oracle refused, the GPU did not' % case)
        got = ma.build_nearfield(*args, x_pts=x, y_pts=y)
        # samples exactly equidistant from two cells (on a mirror line of the hex lattice): the
        # reference takes cKDTree's pick, and so does the HIP path (metalens_amd/ties.py asks
        # cKDTree about exactly those samples) - they are compared like every other sample
        ties_total += int(np.count_nonzero(dec.get('nearest_tie', False)))   # absent: empty window
        for g, w in zip(got[:4], want[:4]):
            err, flips = field_errors(g, w)
            worst = max(worst, err)
            flips_total += flips
            assert flips == 0 and err < 1e-12, (case, err, flips, cx, cy, pol)
        assert abs(got[6] - want[6]) <= 1e-12 * abs(want[6]) or want[6] == 0
    return worst, flips_total, ties_total


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    orders = sys.argv[3] if len(sys.argv) > 3 else 'survey'
    t0 = time.time()
    worst, flips_total, ties_total = run(n_cases, seed, orders)
    print('%s orders: %d cases, worst relative field error %.2e, decision flips %d, exact-tie samples met %d, '
          '%.1f s' % (orders, n_cases, worst, flips_total, ties_total, time.time() - t0))


if __name__ == '__main__':
    main()
