"""Host-side data contract for grating unit cells and collections.

Only what the near-field hot path consumes is kept:
``Grating`` carries periods, ``n_glass`` and the ``characterize()`` records in
``.data``; ``GratingCollection.build_interpolators()`` packs those records into
complex ``[U, V, G+2]`` grids with the reference's zero-fill and +-1 % period
padding rules (reference grating.py:1186-1232) and sets ``.interpolators`` /
``.interpolator_bounds`` with the reference's keys.  Geometry optimisation, the
S4 / Lumerical drivers and plotting are out of scope;
``characterize()`` therefore only accepts records produced elsewhere.

Objects built by the *reference's* own classes work equally well with
metalens_amd.nearfield.build_nearfield - the packer (packing.py) is duck-typed.
"""
import math

import numpy as np

from .interp import TrilinearTable

# glass index the reference's solver scripts use (reference grating.py:1274-1288)
_N_GLASS_TABLE = {450: 1.466, 500: 1.462, 525: 1.461, 550: 1.46, 575: 1.459,
                  580: 1.459, 600: 1.458, 625: 1.457, 650: 1.457}

AMPLITUDE_NAMES = ('ampfy', 'ampfx', 'ampry', 'amprx')


def n_glass(wavelength_in_nm):
    """Tabulated substrate index; ``ValueError`` for an untabulated wavelength
    exactly as the reference does (grating.py:1286-1287)."""
    if wavelength_in_nm not in _N_GLASS_TABLE:
        raise ValueError('bad wavelength' + repr(wavelength_in_nm))
    return _N_GLASS_TABLE[wavelength_in_nm]


class Grating:
    """One periodic unit cell: ``grating_period`` (radial) x ``lateral_period``
    (azimuthal).  ``data`` is the list of ``characterize()`` records, each a dict
    with keys ``wavelength_in_nm, ux, uy, ox, oy, x_or_y, ampfy, ampfx, ampry,
    amprx`` (reference grating.py:453-463).  ``n_glass == 0`` means "use the
    tabulated value" (reference nearfield.py:111-113)."""

    def __init__(self, lateral_period, cyl_height, grating_period=None,
                 target_wavelength=None, angle_in_air=None, n_glass=0, n_tio2=0,
                 xyrra_list_in_nm_deg=None, data=None):
        if grating_period is not None:
            if target_wavelength is not None or angle_in_air is not None:
                raise AssertionError('give grating_period OR (angle_in_air, target_wavelength)')
            self.grating_period = grating_period
        else:
            self.grating_period = target_wavelength / math.sin(angle_in_air)
        self.lateral_period = lateral_period
        self.cyl_height = cyl_height
        self.n_glass = n_glass
        self.n_tio2 = n_tio2
        if xyrra_list_in_nm_deg is not None:
            self.xyrra_list_in_nm_deg = np.array(xyrra_list_in_nm_deg, dtype=float)
        if data is not None:
            self.data = data

    def get_angle_in_air(self, target_wavelength):
        """Incidence angle (in air) at which a lens designed for
        ``target_wavelength`` would place this cell (reference grating.py:195-201)."""
        if self.grating_period < target_wavelength:
            raise ValueError('bad inputs!', target_wavelength, self.grating_period)
        return math.asin(target_wavelength / self.grating_period)

    def characterize(self, records):
        """Attach externally produced x/y-polarisation records (the S4 run that
        produces them is out of scope)."""
        self.data = list(records)


def _axes_and_orders(grating_list):
    ux = sorted({e['ux'] for g in grating_list for e in g.data})
    uy = sorted({e['uy'] for g in grating_list for e in g.data})
    wavelengths = sorted({round(e['wavelength_in_nm']) for g in grating_list for e in g.data})
    orders = sorted({(e['ox'], e['oy']) for g in grating_list for e in g.data})
    return ux, uy, wavelengths, orders


class GratingCollection:
    """Gratings for a range of deflection angles, sorted by ``grating_period``."""

    def __init__(self, target_wavelength, lateral_period, lens_type='cyl',
                 grating_list=None):
        if lens_type not in ('cyl', 'round'):
            raise AssertionError('lens_type must be "cyl" or "round"')
        self.target_wavelength = target_wavelength
        self.lateral_period = lateral_period
        self.lens_type = lens_type
        self.grating_list = list(grating_list) if grating_list is not None else []
        self.grating_list.sort(key=lambda g: g.grating_period)

    def add_one(self, new_grating):
        self.grating_list.append(new_grating)
        self.grating_list.sort(key=lambda g: g.grating_period)

    def get_innermost(self):
        return self.grating_list[-1]

    def get_outermost(self):
        return self.grating_list[0]

    def build_interpolators(self):
        """Pack ``.data`` into one complex grid per
        ``(wavelength_in_nm, (ox, oy), 'x'|'y', 'ampfy'|'ampfx')`` key
        (reference grating.py:1186-1232):

        * axes: sorted distinct ``ux`` and ``uy`` over all records, sorted
          distinct grating periods;
        * a grid node with no record is 0;
        * the period axis gets one extra node at each end, at 0.99*min and
          1.01*max, holding copies of the first / last slab.
        """
        ux, uy, wavelengths, orders = _axes_and_orders(self.grating_list)
        periods = sorted({g.grating_period for g in self.grating_list})
        iu = {u: i for i, u in enumerate(ux)}
        iv = {u: i for i, u in enumerate(uy)}
        ip = {p: i for i, p in enumerate(periods)}
        padded_periods = np.hstack((0.99 * min(periods), periods, 1.01 * max(periods)))
        shape = (len(ux), len(uy), len(periods) + 2)

        slabs = {}
        for g in self.grating_list:
            k = ip[g.grating_period] + 1
            for e in g.data:
                wl = round(e['wavelength_in_nm'])
                for amp in ('ampfy', 'ampfx'):
                    key = (wl, (e['ox'], e['oy']), e['x_or_y'], amp)
                    if key not in slabs:
                        slabs[key] = np.zeros(shape, dtype=complex)
                    # the reference resolves duplicate records "last one wins"
                    # through a dict comprehension (grating.py:1198-1199)
                    slabs[key][iu[e['ux']], iv[e['uy']], k] = e[amp]
        self.interpolators = {}
        for wl in wavelengths:
            for order in orders:
                for pol in ('x', 'y'):
                    for amp in ('ampfy', 'ampfx'):
                        v = slabs.get((wl, order, pol, amp))
                        if v is None:
                            v = np.zeros(shape, dtype=complex)
                        v[:, :, 0] = v[:, :, 1]
                        v[:, :, -1] = v[:, :, -2]
                        self.interpolators[(wl, order, pol, amp)] = TrilinearTable(
                            (ux, uy, padded_periods), v)
        self.interpolator_bounds = (min(ux), max(ux), min(uy), max(uy),
                                    min(padded_periods), max(padded_periods))
