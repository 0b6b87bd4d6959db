"""Lens layout: which grating ring / which centre cell sits where.

This is SURVEY.md §8(f) row 1 - the producer of the two summaries that the
near-field path consumes (``lens_periphery_summary`` dict and
``lens_center_summary`` ``[C,3]`` array).  It restates the reference's
design_collimator.py:57-228,273-313 as array code so that realistic layouts can
be generated where the reference is not available (the GPU box) and for lenses
with ~10^6 centre cells in well under a second.

The CAD exporters (DXF / SVG) are out of scope.
"""
import math
from math import pi

import numpy as np

from . import constants
from .constants import nm, um

#: nearest-neighbour pillar separation of the centre lattice and design
#: wavelength - the reference's module constants (design_collimator.py:34,50,54)
DEFAULT_PITCH = 320 * nm
DEFAULT_WAVELENGTH = 580 * nm
DEFAULT_INDEX_BEFORE_LENS = 1


def target_phase(x, source_distance, wavelength=DEFAULT_WAVELENGTH,
                 refractive_index=DEFAULT_INDEX_BEFORE_LENS):
    """Hyperbolic collimator phase at distance ``x`` from the axis, in
    [0, 2pi) (reference design_collimator.py:57-60).  ``x`` may be an array."""
    k = 2 * pi * refractive_index / wavelength
    return (-k * (np.sqrt(source_distance ** 2 + np.asarray(x, dtype=float) ** 2)
                  - source_distance)) % (2 * pi)


def target_phase_zeros(radius, source_distance, wavelength=DEFAULT_WAVELENGTH,
                       refractive_index=DEFAULT_INDEX_BEFORE_LENS):
    """Radii at which the target phase wraps (Fresnel-zone boundaries), from 0
    up to and including the first one >= ``radius``
    (reference design_collimator.py:62-71)."""
    k = 2 * pi * refractive_index / wavelength
    zeros = []
    order = 0
    while not zeros or zeros[-1] < radius:
        zeros.append((((2 * pi * order) / k + source_distance) ** 2
                      - source_distance ** 2) ** 0.5)
        order += 1
    return zeros


def hexagonal_grid(n, radius):
    """All points of the hexagonal lattice with nearest-neighbour distance
    ``n`` (basis ``(0,n)`` and ``(n*sqrt(3)/2, n/2)``) strictly inside the
    circle of ``radius`` - the reference's ``fourfold_symmetry=False`` case
    (design_collimator.py:75-118), in the same order (column by column)."""
    corner = [(radius, radius), (radius, -radius), (-radius, radius), (-radius, -radius)]
    n1c = [y / n - x / (n * 3 ** 0.5) for x, y in corner]
    n2c = [2 * x / (n * 3 ** 0.5) for x, y in corner]
    n1 = np.arange(int(min(n1c)) - 2, int(max(n1c)) + 3)
    n2 = np.arange(int(min(n2c)) - 2, int(max(n2c)) + 3)
    x = (n * n2) * 3 ** 0.5 / 2
    out = []
    # column-at-a-time keeps peak memory at O(len(n1)) while preserving order
    r2 = radius ** 2
    for n2_here, x_here in zip(n2, x):
        y = n * (n1 + n2_here / 2)
        keep = x_here ** 2 + y ** 2 < r2
        if keep.any():
            col = np.empty((int(keep.sum()), 2))
            col[:, 0] = x_here
            col[:, 1] = y[keep]
            out.append(col)
    if not out:
        return np.zeros((0, 2))
    return np.vstack(out)


def design_center(hgs, source_distance, radius, pitch=DEFAULT_PITCH,
                  wavelength=DEFAULT_WAVELENGTH):
    """``lens_center_summary``: one row ``[x, y, index into hgs.grating_list]``
    per centre cell (reference design_collimator.py:120-137, including its
    ``+ pi`` phase offset that makes centre and periphery add in phase)."""
    xy = hexagonal_grid(pitch, radius)
    if not hasattr(hgs, 'x_amp_list'):
        raise ValueError('Need to run characterize() first')
    r = (xy[:, 0] ** 2 + xy[:, 1] ** 2) ** 0.5
    phase = target_phase(r, source_distance, wavelength) + pi
    x_amp = np.asarray(hgs.x_amp_list)
    best = np.empty(xy.shape[0])
    step = 1 << 16
    for s in range(0, xy.shape[0], step):
        fom = (x_amp[None, :] * np.exp(-1j * phase[s:s + step, None])).imag
        best[s:s + step] = np.argmax(fom, axis=1)
    return np.column_stack((xy, best))


def design_periphery(collections, source_distance, radius,
                     wavelength=None, units=None):
    """``lens_periphery_summary`` for ``collections =
    [[(angle_start, angle_end), GratingCollection], ...]``: one ring per Fresnel
    zone beyond the first switch angle, out to the first ring whose outer edge
    passes ``radius`` (reference design_collimator.py:148-228).  ``units``: the
    caller's unit system (constants.as_units; default SI) - the default wavelength
    and the 2 um of margin are lengths."""
    units = constants.as_units(units)
    um = constants.um * (units.nm / constants.nm)   # (exactly 1e-6 in SI)
    wavelength = 580 * units.nm if wavelength is None else wavelength
    if len(collections) == 0:
        raise AssertionError('need at least one collection')
    for a, b in zip(collections[:-1], collections[1:]):
        if a[0][1] != b[0][0]:
            raise AssertionError('collection angle ranges must abut')
    if not all(c[0][0] < c[0][1] for c in collections):
        raise AssertionError('empty angle range')
    r_switch = source_distance * math.tan(collections[0][0][0])
    zeros = [z for z in target_phase_zeros(radius + 2 * um, source_distance, wavelength)
             if z > r_switch]
    if len(zeros) <= 1:
        raise ValueError('Periphery is too small for even one ring')
    rings = []
    which = 0
    z = 0
    while True:
        r_inner, r_outer = zeros[z], zeros[z + 1]
        r_center = (r_outer + r_inner) / 2
        if collections[which][0][1] < math.atan(r_center / source_distance):
            which += 1
            if which >= len(collections):
                raise ValueError('radius is too big for provided collections')
            continue
        gc = collections[which][1]
        rings.append((r_center, r_outer - r_inner, which,
                      int(round(2 * pi * source_distance / gc.lateral_period))))
        if r_outer > radius:
            break
        z += 1
    r_center = np.array([r[0] for r in rings])
    period = np.array([r[1] for r in rings])
    return {'gratingcollection_list': [c[1] for c in collections],
            'r_center_list': r_center,
            'r_min_list': r_center - 0.5 * period,
            'r_max_list': r_center + 0.5 * period,
            'grating_period_list': period,
            'gratingcollection_index_here_list': np.array([r[2] for r in rings]),
            'num_around_circle_list': np.array([r[3] for r in rings])}


def make_design(collections, source_distance, radius, hgs, pitch=None,
                wavelength=None, units=None):
    """Periphery + centre of a round lens; returns
    ``(lens_periphery_summary, lens_center_summary, r_for_switch)``
    (reference design_collimator.py:273-313; the centre stops 300 nm short of
    the first ring).  ``units``: the caller's unit system (constants.as_units;
    default SI): the defaults - pitch 320 nm, wavelength 580 nm
    (design_collimator.py:34,50) - and the 300 nm are lengths."""
    units = constants.as_units(units)
    nm = units.nm
    pitch = 320 * nm if pitch is None else pitch
    wavelength = 580 * nm if wavelength is None else wavelength
    if len(collections) > 0:
        for _, gc in collections:
            if gc.lens_type != 'round':
                raise AssertionError('round-lens collections only')
            for g in gc.grating_list:
                if (g.n_tio2, g.n_glass, g.cyl_height) != (hgs.n_tio2, hgs.n_glass, hgs.cyl_height):
                    raise AssertionError('centre and periphery materials differ')
        periphery = design_periphery(collections, source_distance, radius, wavelength, units)
        r_for_switch = periphery['r_min_list'][0]
        if not r_for_switch < radius:
            raise AssertionError('no room for the periphery')
    else:
        periphery = None
        r_for_switch = radius
    center = design_center(hgs, source_distance, r_for_switch - 300 * nm, pitch, wavelength)
    return periphery, center, r_for_switch
