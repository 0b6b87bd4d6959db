"""The caller's unit system (``units=``): the reference's users hold lengths, ``c0`` and ``Z0`` in the
units ``numericalunits`` drew for their process (reference nearfield.py:14-15 - never reset), so
``580 * nm`` is NOT 5.8e-7 there.  Everything on the path is covariant under such a change except what
this package derives itself: the table key ``int(round(wavelength / nm))`` (nearfield.py:111), the
defaults (c0, Z0, dipole moment, design lengths) and the nanometres in the period-bound errors."""
import math

import numpy as np
import pytest

import metalens_amd as ma
from metalens_amd import constants, layout, packing, synthetic
from metalens_amd.nearfield import nearfield_params

U = constants.Units(m=3.7, kg=0.21, s=1.9, C=0.6)   # (what numericalunits might have drawn)


def build(units, radius_nm=20000):
    nm = constants.as_units(units).nm
    return synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet), layout.make_design,
                               radius=radius_nm * nm, numerical_aperture=0.3, wavelength=580 * nm,
                               switch_angle=8 * math.pi / 180, num_gratings=8, num_entries=8,
                               design_kwargs={'units': units}, units=units)


def test_units_object():
    si = constants.SI
    assert (si.nm, si.c0, si.Z0) == (constants.nm, constants.c0, constants.Z0)
    assert constants.as_units(None) is si and constants.as_units(U) is U
    with pytest.raises(TypeError):
        constants.as_units(object())
    # derived the way numericalunits derives them: c0 is a speed, Z0 = mu0 c0 an impedance
    assert U.c0 == constants.c0 * 3.7 / 1.9
    assert abs(U.Z0 / (constants.Z0 * 0.21 * 3.7 ** 2 / (1.9 * 0.6 ** 2)) - 1) < 1e-15
    assert constants.default_dipole_moment(U) == 1e-30 * 0.6 * 3.7


def test_same_lens_in_two_unit_systems_packs_alike():
    a, b = build(None), build(U)
    L = U.m
    assert abs(b['source_distance'] / a['source_distance'] - L) < 1e-13 * L
    Sa, Sb = a['lens_periphery_summary'], b['lens_periphery_summary']
    for key in ('r_min_list', 'r_max_list', 'r_center_list', 'grating_period_list'):
        assert np.allclose(np.asarray(Sb[key]) / L, Sa[key], rtol=1e-12, atol=0)
    for key in ('gratingcollection_index_here_list', 'num_around_circle_list'):
        assert np.array_equal(Sa[key], Sb[key])
    assert a['lens_center_summary'].shape == b['lens_center_summary'].shape
    assert np.allclose(b['lens_center_summary'][:, :2] / L, a['lens_center_summary'][:, :2], rtol=0, atol=1e-18)
    assert np.array_equal(a['lens_center_summary'][:, 2], b['lens_center_summary'][:, 2])
    # the table key is the wavelength in NANOMETRES of the caller's system: 580 in both
    wl_a, wl_b = int(round(a['wavelength'] / constants.nm)), int(round(b['wavelength'] / U.nm))
    assert wl_a == wl_b == 580
    assert int(round(b['wavelength'] / constants.nm)) != 580      # (what an SI-only host side would look up)
    for ga, gb in zip(Sa['gratingcollection_list'] + [a['hexgridset']], Sb['gratingcollection_list'] + [b['hexgridset']]):
        ta, tb = packing.pack_table(ga, wl_a), packing.pack_table(gb, wl_b)
        assert np.array_equal(ta['orders'], tb['orders']) and np.array_equal(ta['values'], tb['values'])
        assert np.array_equal(ta['axes'][0], tb['axes'][0]) and np.array_equal(ta['axes'][1], tb['axes'][1])
        if ga is not a['hexgridset']:
            assert np.allclose(tb['axes'][2] / L, ta['axes'][2], rtol=1e-12, atol=0)
            assert np.allclose(tb['bounds'][4:] / L, ta['bounds'][4:], rtol=1e-12, atol=0)
    # the per-call scalars: wavenumbers are inverse lengths, H_coef = c0 k^2 p / 4 pi is a current
    pa = nearfield_params(0.0, 0.0, -a['source_distance'], 'x', a['wavelength'], 1.459,
                          constants.default_dipole_moment(constants.SI), constants.c0, constants.Z0)
    pb = nearfield_params(0.0, 0.0, -b['source_distance'], 'x', b['wavelength'], 1.459,
                          constants.default_dipole_moment(U), U.c0, U.Z0)
    assert abs(pb.kvac * L / pa.kvac - 1) < 1e-13 and abs(pb.k_glass * L / pa.k_glass - 1) < 1e-13
    assert abs(pb.H_coef / (U.C / U.s) / pa.H_coef - 1) < 1e-13


@pytest.mark.gpu
def test_nearfield_and_farfield_are_covariant_on_the_gpu():
    """the same lens and source in SI and in a drawn unit system: H in A/m, E in V/m, the incident
    power in W and the far-field power per solid angle in W/sr agree to rounding once each is
    expressed in the other's units - and a plain SI-only call on the drawn numbers cannot even find
    its table."""
    a, b = build(None), build(U)
    out = {}
    for name, lens, units in (('si', a, None), ('u', b, U)):
        un = constants.as_units(units)
        args = dict(source_x=300 * un.nm, source_y=-200 * un.nm, source_z=-lens['source_distance'], source_pol='x',
                    wavelength=lens['wavelength'], lens_periphery_summary=lens['lens_periphery_summary'],
                    lens_center_summary=lens['lens_center_summary'], hexgridset=lens['hexgridset'])
        nf = ma.build_nearfield(**args, units=units)
        ffts = [np.fft.fft2(np.fft.fftshift(F)) for F in nf[:4]]
        ff = ma.farfield_from_nearfield(*ffts, nf[4], nf[5], lens['wavelength'], nf[7], units=units)
        out[name] = (nf, ff)
    (nfa, ffa), (nfb, ffb) = out['si'], out['u']
    H_unit, E_unit, W = U.C / (U.s * U.m), U.V / U.m, U.kg * U.m ** 2 / U.s ** 3
    assert np.allclose(nfb[4] / U.m, nfa[4], rtol=1e-12, atol=0)
    scale_E = np.abs(nfa[0]).max()
    scale_H = np.abs(nfa[2]).max()
    for k in (0, 1):
        assert np.abs(nfb[k] / E_unit - nfa[k]).max() <= 1e-11 * scale_E
        assert np.abs(nfb[k + 2] / H_unit - nfa[k + 2]).max() <= 1e-11 * scale_H
    assert abs(nfb[6] / W / nfa[6] - 1) < 1e-11                       # incident power through the lens
    assert nfb[7] == nfa[7]
    Pa, Pb = ffa[0], ffb[0] / W                                       # W per unit direction-space area
    ok = np.isfinite(Pa)
    assert np.array_equal(ok, np.isfinite(Pb))
    assert np.abs(Pb[ok] - Pa[ok]).max() <= 1e-10 * Pa[ok].max()
    assert abs(ffb[1] / W / ffa[1] - 1) < 1e-10                        # total_P
    with pytest.raises((KeyError, ValueError)):
        ma.build_nearfield(source_x=0.0, source_y=0.0, source_z=-b['source_distance'], source_pol='x',
                           wavelength=b['wavelength'], lens_periphery_summary=b['lens_periphery_summary'],
                           lens_center_summary=b['lens_center_summary'], hexgridset=b['hexgridset'])
