"""Pin the CPU oracle to fixtures produced by the reference itself
(tests/golden/gen/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import golden_io
from oracle import farfield_oracle, nearfield_oracle, rgi

TOL = 1e-14   # relative to the largest field magnitude of the case (measured: <= 5e-15)

# lens B: the three-order tables; lens C: the orders characterize() would record (eleven / seven per ring
# collection, three in the centre); lens D: a five-order and a three-order collection, one order in the centre
CASES = sorted(os.path.basename(p) for p in glob.glob(golden_io.golden_path('nearfield_[BCD]_*.npz')))


def run_oracle(case, **override):
    lens = golden_io.load_lens(golden_io.golden_path(str(case['lens'])))
    args = dict(source_x=float(case['source_x']), source_y=float(case['source_y']),
                source_z=float(case['source_z']), source_pol=str(case['source_pol']),
                wavelength=float(case['wavelength']), lens_periphery_summary=lens[0],
                lens_center_summary=lens[1], hexgridset=lens[2],
                x_pts=case['x_pts'], y_pts=case['y_pts'],
                dipole_moment=float(case['dipole_moment']), c0=float(case['c0']), Z0=float(case['Z0']))
    args.update(override)
    return nearfield_oracle.build_nearfield(**args)


def rel_err(got, want, scale):
    return np.abs(got - want).max() / scale


@pytest.mark.parametrize('name', CASES)
def test_nearfield_windows(name):
    case = np.load(golden_io.golden_path(name))
    out = run_oracle(case)
    for got, key in zip(out[:4], ('Ex', 'Ey', 'Hx', 'Hy')):
        scale = max(np.abs(case[key]).max(), 1e-300)
        assert got.shape == case[key].shape
        assert rel_err(got, case[key], scale) < TOL, key
        assert np.array_equal(got == 0, case[key] == 0), key + ' support'
    assert abs(out[6] - case['power']) <= 1e-13 * abs(case['power'])
    assert out[7] == case['n_glass']


def test_nearfield_default_grid():
    case = np.load(golden_io.golden_path('nearfield_A_default_grid.npz'))
    out = run_oracle(case, x_pts=None, y_pts=None)
    assert np.array_equal(out[4], case['x_pts']) and np.array_equal(out[5], case['y_pts'])
    s = int(case['stride'])
    sl = (slice(s // 2, None, s), slice(s // 3, None, s))
    for i, key in enumerate(('Ex', 'Ey', 'Hx', 'Hy')):
        scale = case['norms'][i]
        assert rel_err(out[i][sl], case[key], scale) < TOL, key
        assert abs(out[i].sum() - case['sums'][i]) < 1e-11 * scale * np.sqrt(out[i].size), key
        assert abs(np.abs(out[i]).max() - scale) < TOL * scale
        assert (out[i] != 0).sum() == case['nonzero'][i]
    assert abs(out[6] - case['power']) <= 1e-13 * abs(case['power'])


def test_big_equals_single():
    ref = np.load(golden_io.golden_path('nearfield_A_big_vs_single.npz'))
    # the reference itself: strips reproduce the single call exactly
    assert np.all(ref['max_abs_diff'] == 0)
    case = np.load(golden_io.golden_path('nearfield_B_straddle_offaxis_y.npz'))
    lens = golden_io.load_lens(golden_io.golden_path(str(case['lens'])))
    kw = dict(source_x=float(case['source_x']), source_y=float(case['source_y']),
              source_z=float(case['source_z']), source_pol=str(case['source_pol']),
              wavelength=float(case['wavelength']), lens_periphery_summary=lens[0],
              lens_center_summary=lens[1], hexgridset=lens[2], x_pts=case['x_pts'],
              y_pts=case['y_pts'], c0=float(case['c0']), Z0=float(case['Z0']))
    one = nearfield_oracle.build_nearfield(**kw)
    big = nearfield_oracle.build_nearfield_big(pts_at_a_time=48 * 7, **kw)
    for a, b in zip(one[:4], big[:4]):
        assert np.array_equal(a, b)
    assert abs(one[6] - big[6]) <= 1e-14 * abs(one[6])


def test_negative_cases():
    neg = np.load(golden_io.golden_path('negative_cases.npz'))
    base = np.load(golden_io.golden_path('nearfield_B_periphery_onaxis_x.npz'))
    center = np.load(golden_io.golden_path('nearfield_B_center_onaxis_x.npz'))
    um = 1e-6
    wx = base['x_pts']
    overrides = {
        'coarse_pitch': dict(x_pts=wx[::2]),
        'nonuniform': dict(x_pts=np.hstack((wx[:-1], wx[-1] + 1e-9))),
        'source_above': dict(source_z=1e-6),
        'bad_pol': dict(source_pol='s'),
        'plane_z': dict(source_z=-float('inf'), source_pol='z'),
        'ux_overrun': dict(source_x=-180 * um),
        'uy_overrun': dict(source_y=150 * um, source_x=100 * um),
        'plane_wave_overrun': dict(source_z=-float('inf')),
        'center_overrun': dict(source_x=-60 * um, source_z=-60 * um, x_pts=center['x_pts'],
                               y_pts=center['y_pts']),
    }
    for label, ov in overrides.items():
        exc = {'AssertionError': AssertionError, 'ValueError': ValueError}[str(neg[label + '_type'])]
        with pytest.raises(exc) as info:
            run_oracle(base, **ov)
        if exc is ValueError:
            assert info.value.args[0] == str(neg[label + '_msg'])
            np.testing.assert_allclose([float(v) for v in info.value.args[1:]], neg[label + '_vals'],
                                       rtol=1e-12)
    with pytest.raises(ValueError) as info:
        nearfield_oracle.tabulated_n_glass(532)
    assert info.value.args[0] == str(neg['bad_wavelength_msg'])


def test_rgi_semantics():
    z = np.load(golden_io.golden_path('rgi_samples.npz'))
    got = rgi.trilinear((z['axis0'], z['axis1'], z['axis2']), z['values'],
                        z['points'][:, 0], z['points'][:, 1], z['points'][:, 2])
    assert np.array_equal(got, z['result'])   # bit-exact restatement of scipy's arithmetic


def test_good_fft_number():
    z = np.load(golden_io.golden_path('good_fft_number.npz'))
    for g, a in zip(z['goals'], z['answers']):
        assert nearfield_oracle.good_fft_number(g) == a
    with pytest.raises(AssertionError):
        nearfield_oracle.good_fft_number(1e5)


def test_farfield_window():
    z = np.load(golden_io.golden_path('farfield_B_periphery_window.npz'))
    nf = np.load(golden_io.golden_path(str(z['nearfield'])))
    P, total_P, ux, uy, dux, duy = farfield_oracle.farfield_from_nearfield(
        z['fftEx'], z['fftEy'], z['fftHx'], z['fftHy'], nf['x_pts'], nf['y_pts'],
        float(z['wavelength']), float(z['n_glass']), Z0=float(z['Z0']))
    assert np.array_equal(np.isnan(P), np.isnan(z['P']))
    assert np.isnan(P).any()
    ok = ~np.isnan(P)
    assert np.abs(P[ok] - z['P'][ok]).max() <= 1e-14 * np.nanmax(z['P'])
    assert abs(total_P - z['total_P']) <= 1e-14 * abs(z['total_P'])
    assert np.array_equal(ux, z['ux']) and np.array_equal(uy, z['uy'])
    assert dux == z['dux'] and duy == z['duy']


def test_direct_sum_matches_fft_bins():
    """the aperture->direction sum on FFT-lattice directions reproduces the
    reference's N, L (= FFT bin x dA), SURVEY.md §8(c)(iv)"""
    z = np.load(golden_io.golden_path('farfield_B_periphery_window.npz'))
    nf = np.load(golden_io.golden_path(str(z['nearfield'])))
    wl, n = float(z['wavelength']), float(z['n_glass'])
    x, y = nf['x_pts'], nf['y_pts']
    ux = farfield_oracle.fft_direction_cosines(len(x), x[1] - x[0], wl, n)
    uy = farfield_oracle.fft_direction_cosines(len(y), y[1] - y[0], wl, n)
    got = farfield_oracle.radiation_vectors(nf['Ex'], nf['Ey'], nf['Hx'], nf['Hy'], x, y, wl, n, ux, uy)
    for g, key in zip(got, ('Nx', 'Ny', 'Lx', 'Ly')):
        assert np.abs(g - z[key]).max() <= 1e-13 * np.abs(z[key]).max(), key
    # and the pair-list form agrees with the tensor form
    ii = np.array([0, 3, 17, 39]); jj = np.array([0, 5, 40, 47])
    pairs = farfield_oracle.radiation_vectors_pairs(nf['Ex'], nf['Ey'], nf['Hx'], nf['Hy'], x, y,
                                                    wl, n, ux[ii], uy[jj])
    for g, key in zip(pairs, ('Nx', 'Ny', 'Lx', 'Ly')):
        assert np.abs(g - z[key][ii, jj]).max() <= 1e-13 * np.abs(z[key]).max(), key


def test_farfield_lens_A_lattice():
    z = np.load(golden_io.golden_path('farfield_A_lattice.npz'))
    case = np.load(golden_io.golden_path(str(z['nearfield'])))
    out = run_oracle(case, x_pts=None, y_pts=None)
    ffts = [np.fft.fft2(np.fft.fftshift(F)) for F in out[:4]]
    P, total_P, ux, uy, dux, duy = farfield_oracle.farfield_from_nearfield(
        *ffts, out[4], out[5], float(z['wavelength']), float(z['n_glass']), Z0=float(z['Z0']))
    s = int(z['stride'])
    sub = P[3::s, 2::s]
    ok = ~np.isnan(z['P'])
    assert np.array_equal(np.isnan(sub), ~ok)
    assert np.abs(sub[ok] - z['P'][ok]).max() <= 1e-11 * z['P_max']
    assert np.isnan(P).sum() == z['P_nan_count']
    assert abs(total_P - z['total_P']) <= 1e-11 * abs(z['total_P'])
    assert tuple(np.unravel_index(np.nanargmax(P), P.shape)) == tuple(z['P_argmax'])


def test_tie_breaker_names_the_fixtures_scipy():
    """ties.settle asks scipy's cKDTree which of two equidistant cells wins (the reference's own choice,
    nearfield.py:363-364); the version the fixtures were made with is recorded in every .npz and must be
    the one ties.py warns against departing from"""
    import glob
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        '_ties_src', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'metalens_amd', 'ties.py'))
    src = open(spec.origin).read()
    here = os.path.dirname(os.path.abspath(__file__))
    versions = set()
    for f in glob.glob(os.path.join(here, 'golden', '*.npz')):
        z = np.load(f, allow_pickle=True)
        if 'scipy' in z.files:
            versions.add(str(z['scipy']))
    assert len(versions) == 1
    assert "FIXTURE_SCIPY = '%s'" % versions.pop() in src
