"""Binary persistence of characterisation data and packed interpolation grids
(SURVEY.md §8(f) row 2).

The reference persists a characterised ``GratingCollection`` / ``HexGridSet`` by printing
Python source (``repr()``, grating.py:263-281,1082-1093; lens_center.py:59-78; README.md:29-34):
multi-megabyte text that has to be pasted back and re-parsed, and the interpolators are rebuilt
afterwards.  This module writes one ``.npz`` per object instead: the ``characterize()`` records as
columnar arrays plus, if present, the packed grids exactly as ``build_interpolators()`` produced
them (zero-filled holes and +-1 % period padding included), so that loading gives an object that
is immediately usable by ``build_nearfield``.

Format (version 1), all arrays little-endian as NumPy writes them:
  kind                 'GratingCollection' | 'HexGridSet'
  meta                 float64: GC [target_wavelength, lateral_period]; HGS [sep, cyl_height,
                       n_glass, n_tio2]
  lens_type            GC only
  g_geometry[n,5]      per grating: lateral_period, cyl_height, grating_period, n_glass, n_tio2
  g_offsets[n+1]       record ranges per grating into the rec_* columns
  rec_wavelength_in_nm, rec_ux, rec_uy (float64); rec_ox, rec_oy (int32); rec_pol (uint8, 0='x');
  rec_amps[:,4]        complex128: ampfy, ampfx, ampry, amprx
  x_amp_list           HGS only
  interp_keys / interp_axis0..2 / interp_values / interp_bounds     (optional) packed grids
"""
import numpy as np

from .grating import Grating, GratingCollection
from .interp import TrilinearTable
from .lens_center import HexGridSet

FORMAT_VERSION = 1
_AMPS = ('ampfy', 'ampfx', 'ampry', 'amprx')


def _records_to_columns(grating_list):
    offsets = [0]
    cols = {k: [] for k in ('wl', 'ux', 'uy', 'ox', 'oy', 'pol')}
    amps = []
    geometry = []
    for g in grating_list:
        data = getattr(g, 'data', [])
        for e in data:
            cols['wl'].append(e['wavelength_in_nm'])
            cols['ux'].append(e['ux'])
            cols['uy'].append(e['uy'])
            cols['ox'].append(e['ox'])
            cols['oy'].append(e['oy'])
            cols['pol'].append(0 if e['x_or_y'] == 'x' else 1)
            amps.append([e[a] for a in _AMPS])
        offsets.append(offsets[-1] + len(data))
        geometry.append([g.lateral_period, g.cyl_height, g.grating_period, g.n_glass, g.n_tio2])
    return {'g_geometry': np.array(geometry, dtype=float).reshape(-1, 5),
            'g_offsets': np.array(offsets, dtype=np.int64),
            'rec_wavelength_in_nm': np.array(cols['wl'], dtype=float),
            'rec_ux': np.array(cols['ux'], dtype=float), 'rec_uy': np.array(cols['uy'], dtype=float),
            'rec_ox': np.array(cols['ox'], dtype=np.int32), 'rec_oy': np.array(cols['oy'], dtype=np.int32),
            'rec_pol': np.array(cols['pol'], dtype=np.uint8),
            'rec_amps': np.array(amps, dtype=complex).reshape(-1, 4)}


def _columns_to_gratings(z):
    gratings = []
    off = z['g_offsets']
    for i, (lat, cyl, per, ng, nt) in enumerate(z['g_geometry']):
        recs = []
        for r in range(int(off[i]), int(off[i + 1])):
            e = {'wavelength_in_nm': float(z['rec_wavelength_in_nm'][r]), 'ux': float(z['rec_ux'][r]),
                 'uy': float(z['rec_uy'][r]), 'ox': int(z['rec_ox'][r]), 'oy': int(z['rec_oy'][r]),
                 'x_or_y': 'x' if z['rec_pol'][r] == 0 else 'y'}
            for k, a in enumerate(_AMPS):
                e[a] = complex(z['rec_amps'][r, k])
            recs.append(e)
        gratings.append(Grating(lateral_period=float(lat), cyl_height=float(cyl),
                                grating_period=float(per),
                                n_glass=(0 if ng == 0 else float(ng)),
                                n_tio2=(0 if nt == 0 else float(nt)), data=recs))
    return gratings


def _pack_interpolators(obj, out):
    if not hasattr(obj, 'interpolators'):
        return
    keys = sorted(obj.interpolators)
    f0 = obj.interpolators[keys[0]]
    for ax in range(3):
        out['interp_axis%d' % ax] = np.asarray(f0.grid[ax], dtype=float)
    out['interp_keys'] = np.array(['%d|%d|%d|%s|%s' % (k[0], k[1][0], k[1][1], k[2], k[3])
                                   for k in keys])
    out['interp_values'] = np.stack([np.asarray(obj.interpolators[k].values) for k in keys])
    out['interp_bounds'] = np.array(obj.interpolator_bounds, dtype=float)


def _unpack_interpolators(obj, z):
    if 'interp_keys' not in z:
        return
    grid = tuple(z['interp_axis%d' % ax] for ax in range(3))
    obj.interpolators = {}
    for s, v in zip(z['interp_keys'], z['interp_values']):
        wl, ox, oy, pol, amp = str(s).split('|')
        obj.interpolators[(int(wl), (int(ox), int(oy)), pol, amp)] = TrilinearTable(grid, v)
    obj.interpolator_bounds = tuple(float(b) for b in z['interp_bounds'])


def save(path, obj):
    """write a GratingCollection or HexGridSet (this package's or the reference's - duck-typed)"""
    out = {'format_version': np.array(FORMAT_VERSION)}
    out.update(_records_to_columns(obj.grating_list))
    if hasattr(obj, 'sep'):
        out['kind'] = np.array('HexGridSet')
        out['meta'] = np.array([obj.sep, obj.cyl_height, obj.n_glass, obj.n_tio2], dtype=float)
        if hasattr(obj, 'x_amp_list'):
            out['x_amp_list'] = np.asarray(obj.x_amp_list, dtype=complex)
    else:
        out['kind'] = np.array('GratingCollection')
        out['meta'] = np.array([obj.target_wavelength, obj.lateral_period], dtype=float)
        out['lens_type'] = np.array(obj.lens_type)
    _pack_interpolators(obj, out)
    np.savez_compressed(path, **out)


def load(path):
    z = np.load(path)
    if int(z['format_version']) != FORMAT_VERSION:
        raise ValueError('unsupported table file version %d' % int(z['format_version']))
    gratings = _columns_to_gratings(z)
    if str(z['kind']) == 'HexGridSet':
        sep, cyl, ng, nt = z['meta']
        obj = HexGridSet(sep=float(sep), cyl_height=float(cyl), n_glass=(0 if ng == 0 else float(ng)),
                         n_tio2=(0 if nt == 0 else float(nt)), grating_list=gratings,
                         x_amp_list=z['x_amp_list'] if 'x_amp_list' in z else None)
    elif str(z['kind']) == 'GratingCollection':
        obj = GratingCollection(target_wavelength=float(z['meta'][0]),
                                lateral_period=float(z['meta'][1]), lens_type=str(z['lens_type']),
                                grating_list=gratings)
    else:
        raise ValueError('unknown object kind %r' % str(z['kind']))
    _unpack_interpolators(obj, z)
    return obj
