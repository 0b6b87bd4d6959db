"""Host-side data contract for the lens centre: a set of hexagonal-lattice unit
cells (one per pillar diameter) whose characterisation tables are indexed by the
integer position of the cell in ``grating_list``.

Only the part the near-field hot path consumes is kept:
``HexGridSet.build_interpolators()`` (reference lens_center.py:188-226) and
``pick_from_phase`` (reference lens_center.py:175-186), which the layout
generator needs.  Running S4 is out of scope.
"""
import numpy as np

from .grating import AMPLITUDE_NAMES, _axes_and_orders
from .interp import TrilinearTable


class HexGridSet:
    def __init__(self, sep, cyl_height, n_glass=0, n_tio2=0, grating_list=None,
                 x_amp_list=None):
        self.sep = sep
        self.nnn_sep = sep * 3 ** 0.5
        self.cyl_height = cyl_height
        self.n_glass = n_glass
        self.n_tio2 = n_tio2
        self.grating_list = list(grating_list) if grating_list is not None else []
        if x_amp_list is not None:
            self.x_amp_list = np.array(x_amp_list)

    def pick_from_phase(self, target_phase):
        """Index of the cell whose normal-incidence transmission phase is the
        best match to ``target_phase`` (reference lens_center.py:175-186)."""
        if not hasattr(self, 'x_amp_list'):
            raise ValueError('Need to run characterize() first')
        return int(np.argmax((self.x_amp_list * np.exp(-1j * target_phase)).imag))

    def build_interpolators(self):
        """One complex ``[U, V, K]`` grid per
        ``(wavelength_in_nm, (ox, oy), 'x'|'y', amplitude name)``; third axis is
        the cell index 0..K-1; missing records are 0; no padding
        (reference lens_center.py:188-226)."""
        if not hasattr(self, 'x_amp_list'):
            raise ValueError('Need to run characterize() first')
        ux, uy, wavelengths, orders = _axes_and_orders(self.grating_list)
        iu = {u: i for i, u in enumerate(ux)}
        iv = {u: i for i, u in enumerate(uy)}
        index_axis = np.arange(len(self.grating_list))
        shape = (len(ux), len(uy), len(self.grating_list))
        slabs = {}
        for k, g in enumerate(self.grating_list):
            for e in g.data:
                for amp in AMPLITUDE_NAMES:
                    # the reference matches wavelength_in_nm by == against the
                    # rounded value (lens_center.py:211-212)
                    if e['wavelength_in_nm'] != round(e['wavelength_in_nm']):
                        continue
                    key = (round(e['wavelength_in_nm']), (e['ox'], e['oy']), e['x_or_y'], amp)
                    if key not in slabs:
                        slabs[key] = np.zeros(shape, dtype=complex)
                        slabs[key + ('n',)] = np.zeros(shape, dtype=int)
                    slabs[key][iu[e['ux']], iv[e['uy']], k] = e[amp]
                    slabs[key + ('n',)][iu[e['ux']], iv[e['uy']], k] += 1
        self.interpolators = {}
        for wl in wavelengths:
            for order in orders:
                for pol in ('x', 'y'):
                    for amp in AMPLITUDE_NAMES:
                        key = (wl, order, pol, amp)
                        v = slabs.get(key)
                        if v is None:
                            v = np.zeros(shape, dtype=complex)
                        elif slabs[key + ('n',)].max() > 1:
                            raise AssertionError('duplicate characterisation record')
                        self.interpolators[key] = TrilinearTable((ux, uy, index_axis), v)
        self.interpolator_bounds = (min(ux), max(ux), min(uy), max(uy),
                                    int(index_axis.min()), int(index_axis.max()))
