"""Synthetic characterisation tables and lenses.

``characterize()`` needs the external S4 solver, so every
test, fixture and benchmark in this repo feeds the near-field path with smooth,
analytic complex amplitudes written into ``Grating.data`` in exactly the record
format the reference parses out of S4 (reference grating.py:453-463).

The builders take the classes to instantiate as arguments so that the very same
records can be put into the reference's own ``Grating`` / ``GratingCollection`` /
``HexGridSet`` (golden-fixture generation, tests/golden/gen/make_golden.py) or
into this package's containers (everything that runs on the GPU box).
"""
import math
from math import pi

import numpy as np

from . import constants
from .constants import degree

PERIPHERY_ORDERS = ((0, 0), (-1, 0), (1, 0))
CENTER_ORDERS = ((0, 0), (-1, 0), (1, 0))


def _smooth_amp(ux, uy, s, ox, oy, pol, amp, seed):
    """A smooth complex function of incidence direction and a shape parameter
    ``s`` in [0,1]; different (but deterministic) for every order / incident
    polarisation / amplitude name."""
    tag = (3 * (ox + 2) + 7 * (oy + 2) + (11 if pol == 'y' else 0)
           + {'ampfy': 0, 'ampfx': 5, 'ampry': 13, 'amprx': 17}[amp] + seed)
    a1 = 0.37 + 0.05 * (tag % 7)
    a2 = 0.21 + 0.04 * (tag % 5)
    a3 = 0.53 + 0.03 * (tag % 11)
    # co-polarised design order is strong, everything else is weak
    co = (pol == 'x' and amp == 'ampfx') or (pol == 'y' and amp == 'ampfy')
    base = 0.8 if (co and (ox, oy) == (-1, 0)) else (0.25 if co else 0.08)
    if amp in ('ampry', 'amprx'):
        base *= 0.3
    mag = base * (1 + 0.2 * math.cos(a1 * 3 * ux + 0.3 * tag) + 0.15 * math.sin(a2 * 4 * uy + s)
                  + 0.1 * math.cos(2 * pi * a3 * s))
    phase = 0.6 * math.sin(a1 * 2 * ux + a2 * uy * 3) + 0.8 * a3 * s + 0.25 * (tag % 3) * ux * uy
    return complex(mag * math.cos(phase), mag * math.sin(phase))


def propagating_orders(ux, uy, wavelength, grating_period, lateral_period, search=5):
    """The orders characterize() records at one direction of incidence: every (ox, oy) in
    [-search, search]^2 whose diffracted wave propagates in air,
    (kx + ox Gx)^2 + (ky + oy Gy)^2 < kvac^2 (reference grating.lua:406-423, strict)."""
    gx, gy = wavelength / grating_period, wavelength / lateral_period
    return [(ox, oy) for ox in range(-search, search + 1) for oy in range(-search, search + 1)
            if (ux + ox * gx) ** 2 + (uy + oy * gy) ** 2 < 1]


def _records(ux_axis, uy_axis, s, orders, wavelength_in_nm, seed, phase0=0.0, periods=None, nm=constants.nm):
    """``orders`` = a fixed list, or 'physical': per direction the orders that propagate there for
    the grating's ``periods`` = (grating_period, lateral_period) - what S4 would have been asked
    for; the table packer zero-fills the rest (reference grating.py:1207-1214)"""
    rot = complex(math.cos(phase0), math.sin(phase0))
    recs = []
    for ux in ux_axis:
        for uy in uy_axis:
            if ux ** 2 + uy ** 2 >= 1:   # (grating.lua:398: no incident wave there)
                continue
            here = orders
            if isinstance(orders, str):
                assert orders == 'physical'
                here = propagating_orders(ux, uy, wavelength_in_nm * nm, *periods)
            for ox, oy in here:
                for pol in ('x', 'y'):
                    e = {'wavelength_in_nm': float(wavelength_in_nm), 'ux': float(ux),
                         'uy': float(uy), 'ox': ox, 'oy': oy, 'x_or_y': pol}
                    for amp in ('ampfy', 'ampfx', 'ampry', 'amprx'):
                        e[amp] = _smooth_amp(ux, uy, s, ox, oy, pol, amp, seed) * rot
                    recs.append(e)
    return recs


def _warp_axis(axis, warp):
    """``warp`` != 0: move the interior nodes off the uniform spacing (ends fixed, order kept)"""
    if not warp:
        return axis
    lo, hi = axis[0], axis[-1]
    t = (axis - lo) / (hi - lo)
    return lo + (hi - lo) * (t + warp * t * (1 - t))


def make_collection(Grating, GratingCollection, angle_lo, angle_hi, target_wavelength,
                    cyl_height=None, n_glass=0, n_tio2=0, num_gratings=24, u_steps=5,
                    local_lateral_period=None, orders=PERIPHERY_ORDERS, seed=0,
                    drop_every=0, axis_warp=0.0, units=None):
    """A 'round'-lens GratingCollection covering incidence angles
    ``[angle_lo, angle_hi]`` (radians) with ``num_gratings`` periods.

    ``drop_every`` > 0 removes every n-th record to exercise the table packer's
    zero-fill rule (reference grating.py:1207-1214).  ``units``: the unit system the lengths are in
    (constants.as_units; default SI); the default pillar height is 550 nm, the default lateral period 330 nm."""
    nm = constants.as_units(units).nm
    cyl_height = 550 * nm if cyl_height is None else cyl_height
    local_lateral_period = 330 * nm if local_lateral_period is None else local_lateral_period
    wl_nm = int(round(target_wavelength / nm))
    margin = 0.6 * degree
    p_min = target_wavelength / math.sin(min(angle_hi + margin, 89 * degree))
    p_max = target_wavelength / math.sin(max(angle_lo - margin, 1 * degree))
    mid = 0.5 * (angle_lo + angle_hi)
    L0 = local_lateral_period / math.tan(mid)
    ux_lo = max(-0.99, math.sin(angle_lo) - 0.25)
    ux_hi = min(0.99, math.sin(angle_hi) + 0.25)
    ux_axis = _warp_axis(np.linspace(ux_lo, ux_hi, u_steps), axis_warp)
    uy_axis = _warp_axis(np.linspace(-0.2, 0.2, u_steps), axis_warp)
    gratings = []
    for i, period in enumerate(np.linspace(p_min, p_max, num_gratings)):
        period = float(period)
        angle = math.asin(target_wavelength / period)
        recs = _records(ux_axis, uy_axis, i / max(1, num_gratings - 1), orders, wl_nm, seed,
                        periods=(period, L0 * math.tan(angle)), nm=nm)
        if drop_every:
            recs = [r for j, r in enumerate(recs) if (j + i) % drop_every != drop_every - 1]
        gratings.append(Grating(lateral_period=L0 * math.tan(angle), cyl_height=cyl_height,
                                grating_period=period, n_glass=n_glass, n_tio2=n_tio2,
                                data=recs))
    return GratingCollection(target_wavelength=target_wavelength, lateral_period=L0,
                             lens_type='round', grating_list=gratings)


def make_hexgridset(Grating, HexGridSet, wavelength, sep=None, cyl_height=None,
                    n_glass=0, n_tio2=0, num_entries=12, u_steps=5, orders=CENTER_ORDERS,
                    seed=100, axis_warp=0.0, units=None):
    """A HexGridSet of ``num_entries`` cells whose normal-incidence phase
    sweeps 0..2pi, characterised over ux,uy in [-0.499, 0.501]
    (reference lens_center.py:88-90).  ``units`` as for ``make_collection``; default separation 320 nm."""
    nm = constants.as_units(units).nm
    sep = 320 * nm if sep is None else sep
    cyl_height = 550 * nm if cyl_height is None else cyl_height
    wl_nm = int(round(wavelength / nm))
    axis = _warp_axis(np.linspace(-0.499, 0.501, u_steps), axis_warp)
    gratings = []
    x_amp = []
    for k in range(num_entries):
        phase0 = 2 * pi * k / num_entries
        recs = _records(axis, axis, k / max(1, num_entries - 1), orders, wl_nm, seed, phase0,
                        periods=(sep * 3 ** 0.5, sep), nm=nm)
        gratings.append(Grating(grating_period=sep * 3 ** 0.5, lateral_period=sep,
                                cyl_height=cyl_height, n_glass=n_glass, n_tio2=n_tio2,
                                data=recs))
        x_amp.append(_smooth_amp(0.001, 0.001, k / max(1, num_entries - 1), 0, 0, 'x', 'ampfx', seed)
                     * complex(math.cos(phase0), math.sin(phase0)))
    return HexGridSet(sep=sep, cyl_height=cyl_height, n_glass=n_glass, n_tio2=n_tio2,
                      grating_list=gratings, x_amp_list=x_amp)


def make_lens(classes, make_design, radius, numerical_aperture, wavelength=None,
              switch_angle=12 * degree, n_glass=0, num_gratings=24, num_entries=12,
              max_collection_span=9 * degree, design_kwargs=None, axis_warp=0.0, u_steps=5,
              periphery_orders=PERIPHERY_ORDERS, center_orders=CENTER_ORDERS, units=None):
    """A complete synthetic round lens of ``radius`` and ``numerical_aperture``
    for an on-axis source at the focal distance ``radius / tan(asin(NA))``.

    ``classes`` = ``(Grating, GratingCollection, HexGridSet)``;
    ``make_design`` = this package's ``layout.make_design`` or the reference's.
    ``periphery_orders`` / ``center_orders``: a list of (ox, oy), or 'physical' - the orders
    characterize() would record (``propagating_orders``), per collection and direction;
    ``periphery_orders`` may also be a list of such lists, one per collection (cycled).
    ``units``: the unit system ``radius`` and ``wavelength`` are in (constants.as_units; default
    SI; ``make_design`` then needs its own ``units`` in ``design_kwargs``).
    Returns a dict with the design inputs and both summaries.
    """
    Grating, GratingCollection, HexGridSet = classes
    nm = constants.as_units(units).nm
    wavelength = 580 * nm if wavelength is None else wavelength
    source_distance = radius / math.tan(math.asin(numerical_aperture))
    # the outermost ring is the first whose outer edge passes `radius`
    edge_angle = math.atan((radius + 4e-6 * (nm / constants.nm)) / source_distance)   # (+ 4 um)
    n_col = max(1, int(math.ceil((edge_angle - switch_angle) / max_collection_span)))
    bounds = np.linspace(switch_angle, edge_angle + 0.2 * degree, n_col + 1)
    collections = []
    per_collection = (not isinstance(periphery_orders, str)
                      and isinstance(periphery_orders[0][0], (tuple, list)))
    for i in range(n_col):
        lo, hi = float(bounds[i]), float(bounds[i + 1])
        gc = make_collection(Grating, GratingCollection, lo, hi, wavelength, n_glass=n_glass,
                             num_gratings=num_gratings, seed=i, axis_warp=axis_warp,
                             u_steps=u_steps, units=units,
                             orders=periphery_orders[i % len(periphery_orders)] if per_collection
                             else periphery_orders)
        collections.append([(lo, hi), gc])
    hgs = make_hexgridset(Grating, HexGridSet, wavelength, n_glass=n_glass,
                          num_entries=num_entries, axis_warp=axis_warp, u_steps=u_steps,
                          orders=center_orders, units=units)
    for _, gc in collections:
        gc.build_interpolators()
    hgs.build_interpolators()
    periphery, center, r_switch = make_design(collections, source_distance, radius, hgs,
                                              **(design_kwargs or {}))
    return {'collections': collections, 'hexgridset': hgs, 'source_distance': source_distance,
            'lens_periphery_summary': periphery, 'lens_center_summary': center,
            'r_for_switch': r_switch, 'wavelength': wavelength, 'radius': radius}
