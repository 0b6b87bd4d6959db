"""Fixture (de)serialisation shared by the golden generator and the tests.

A *lens fixture* holds the inputs of the near-field path as plain arrays: ring
tables, centre cells, and every interpolation grid (axes + complex values) of
every GratingCollection and of the HexGridSet, keyed the way the reference keys
``.interpolators``.  ``load_lens`` rebuilds duck-typed objects exposing exactly
the attributes the reference's ``build_nearfield`` touches:
``.grating_list[i].data`` (only ``ox``/``oy`` are read), ``.grating_list[0]
.n_glass/.grating_period/.lateral_period``, ``.interpolators[key]`` with
``.grid``/``.values`` and ``__call__``, ``.interpolator_bounds``.
"""
import os
from types import SimpleNamespace

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

RING_KEYS = ('r_min_list', 'r_max_list', 'r_center_list', 'grating_period_list',
             'gratingcollection_index_here_list', 'num_around_circle_list')


def _pack_tables(prefix, obj, out):
    keys = sorted(obj.interpolators.keys())
    f0 = obj.interpolators[keys[0]]
    for ax in range(3):
        out['%s_axis%d' % (prefix, ax)] = np.asarray(f0.grid[ax], dtype=float)
    out[prefix + '_keys'] = np.array(['%d|%d|%d|%s|%s' % (k[0], k[1][0], k[1][1], k[2], k[3])
                                     for k in keys])
    out[prefix + '_values'] = np.stack([np.asarray(obj.interpolators[k].values) for k in keys])
    out[prefix + '_bounds'] = np.array(obj.interpolator_bounds, dtype=float)
    g0 = obj.grating_list[0]
    out[prefix + '_g0'] = np.array([g0.n_glass, g0.grating_period, g0.lateral_period], dtype=float)
    out[prefix + '_orders'] = np.array(sorted({(e['ox'], e['oy']) for g in obj.grating_list
                                               for e in g.data}), dtype=int).reshape(-1, 2)


def pack_lens(periphery, center, hgs):
    out = {k: np.asarray(periphery[k]) for k in RING_KEYS}
    out['num_collections'] = np.array(len(periphery['gratingcollection_list']))
    for i, gc in enumerate(periphery['gratingcollection_list']):
        _pack_tables('gc%d' % i, gc, out)
    _pack_tables('hgs', hgs, out)
    out['lens_center_summary'] = np.asarray(center, dtype=float)
    return out


def _unpack_tables(prefix, z):
    from metalens_amd.interp import TrilinearTable
    grid = tuple(z['%s_axis%d' % (prefix, ax)] for ax in range(3))
    interpolators = {}
    for s, v in zip(z[prefix + '_keys'], z[prefix + '_values']):
        wl, ox, oy, pol, amp = str(s).split('|')
        interpolators[(int(wl), (int(ox), int(oy)), pol, amp)] = TrilinearTable(grid, v)
    n_glass, gp, lp = z[prefix + '_g0']
    data = [{'ox': int(o[0]), 'oy': int(o[1])} for o in z[prefix + '_orders']]
    g0 = SimpleNamespace(n_glass=(0 if n_glass == 0 else float(n_glass)), grating_period=float(gp),
                         lateral_period=float(lp), data=data)
    b = z[prefix + '_bounds']
    bounds = tuple(float(x) for x in b)
    return SimpleNamespace(grating_list=[g0], interpolators=interpolators,
                           interpolator_bounds=bounds)


def load_lens(path_or_npz):
    z = np.load(path_or_npz) if isinstance(path_or_npz, str) else path_or_npz
    periphery = {k: z[k] for k in RING_KEYS}
    periphery['gratingcollection_list'] = [_unpack_tables('gc%d' % i, z)
                                           for i in range(int(z['num_collections']))]
    hgs = _unpack_tables('hgs', z)
    return periphery, z['lens_center_summary'], hgs


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name)
