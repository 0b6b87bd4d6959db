#!/usr/bin/env python3
"""Gate A of the fused synthesis -> row transform path, answered from the geometry: how many
distinct RINGS - the (ring, table cell) blocks a wave has to match and stage; the table cell of a ring hardly ever
changes inside a wave - and how many distinct gratings (ring, rotation sector: one rotation-table entry each)
do the 64 samples of a wave span when the wave is an 8 x 8 patch (what the ring kernel runs), 4 x 16, 2 x 32 or a
1 x 64 piece of ONE aperture row (what a kernel that synthesises rows next to the row transform's LDS buffer needs)?
CPU only (the layout restatement + numpy).

    python tools/wave_shape_stats.py [--aperture 4096] [--diameter 1e-3] [--na 0.5]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--aperture', type=int, default=4096)
    ap.add_argument('--diameter', type=float, default=1e-3)
    ap.add_argument('--na', type=float, default=0.5)
    ap.add_argument('--slots', type=int, default=8, help='blocks a round of the ring kernel stages (narrow instantiation: 8)')
    args = ap.parse_args()
    import bench
    wl = 580e-9
    lens, x, u = bench.build_workload(args.aperture, 512, args.diameter, args.na, wl, 1.0)
    S = lens['lens_periphery_summary']
    r_min = np.asarray(S['r_min_list'], dtype=float)
    r_max = np.asarray(S['r_max_list'], dtype=float)
    nac = np.asarray(S['num_around_circle_list'], dtype=np.int64)
    N = x.size
    X, Y = np.meshgrid(x, x, indexing='ij')          # x = axis 0, y fastest (nearfield.py:117)
    r = np.hypot(X, Y)
    ring = np.searchsorted(np.concatenate((r_min, r_max[-1:])), r) - 1      # nearfield.py:125-126
    ring[r > r_max[-1]] = -1
    peri = ring >= 0
    sector = np.zeros_like(ring)
    phi = np.arctan2(Y, X)
    d = 2 * np.pi / nac[np.maximum(ring, 0)]
    sector[peri] = np.rint(phi[peri] / d[peri]).astype(np.int64)             # nearfield.py:169
    key = np.where(peri, ring.astype(np.int64) * (1 << 20) + (sector + (1 << 19)), -1)
    out = {'aperture': N, 'rings': int(r_min.size), 'ring_samples': int(peri.sum())}
    for h, w in ((8, 8), (4, 16), (2, 32), (1, 64)):
        k = key.reshape(N // h, h, N // w, w).transpose(0, 2, 1, 3).reshape(-1, h * w)
        rg = np.where(k >= 0, k >> 20, -1)
        k = np.sort(k, axis=1)
        rg = np.sort(rg, axis=1)
        has = (k >= 0).any(axis=1)
        n_blocks = ((k[:, 1:] != k[:, :-1]) & (k[:, 1:] >= 0)).sum(axis=1) + (k[:, 0] >= 0)
        n_rings = ((rg[:, 1:] != rg[:, :-1]) & (rg[:, 1:] >= 0)).sum(axis=1) + (rg[:, 0] >= 0)
        ng, nb = n_blocks[has], n_rings[has]
        rounds = np.ceil(nb / args.slots)
        out['%dx%d' % (h, w)] = {
            'waves_with_ring_samples': int(has.sum()),
            'blocks(rings)_per_wave_mean': round(float(nb.mean()), 2), 'blocks_p99': int(np.quantile(nb, 0.99)), 'blocks_max': int(nb.max()),
            'gratings_per_wave_mean': round(float(ng.mean()), 2), 'gratings_max': int(ng.max()),
            'waves_needing_a_second_round_%': round(100 * float((rounds > 1).mean()), 2),
            'rounds_per_wave_mean': round(float(rounds.mean()), 3),
            'staged_KB_per_wave_mean(3 orders)': round(float(nb.mean()) * 3 * 16 * 16 / 1024, 2)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
