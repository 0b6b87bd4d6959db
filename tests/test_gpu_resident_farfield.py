"""``farfield_from_resident_nearfield``: the reference's README flow
``build_nearfield -> fft2(fftshift) x 4 -> farfield_from_nearfield`` (README.md:27,
nearfield_farfield.py:14-75) with the near field staying on the GPU.  Pinned on the reference's
own outputs (tests/golden/farfield_*.npz) and on the oracle's restatement of the same flow.
Needs an MI355X."""
import os
import sys

import math

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _upload(ctx, fields):
    from metalens_amd import _lib
    arrs = [_lib.c128(a) for a in fields]
    _lib.check(ctx.lib.ml_fields_upload(ctx.handle, arrs[0].shape[0], arrs[0].shape[1],
                                        *[_lib.dptr(a) for a in arrs]))


def test_reference_lattice_golden_window():
    """the reference's (P, total_P, ux, uy, dux, duy) for a 40 x 48 window, from the window's near
    field resident on the GPU (the GEMM route: 40 and 48 are not multiples of 256)"""
    import metalens_amd as ma
    from metalens_amd import _lib
    z = np.load(golden_io.golden_path('farfield_B_periphery_window.npz'))
    nf = np.load(golden_io.golden_path(str(z['nearfield'])))
    ctx = _lib.default_context()
    _upload(ctx, [nf[k] for k in ('Ex', 'Ey', 'Hx', 'Hy')])
    P, total_P, ux, uy, dux, duy = ma.farfield_from_resident_nearfield(
        nf['x_pts'], nf['y_pts'], float(z['wavelength']), float(z['n_glass']), Z0=float(z['Z0']), ctx=ctx)
    assert np.array_equal(np.isnan(P), np.isnan(z['P'])) and np.isnan(P).any()
    ok = ~np.isnan(P)
    assert np.abs(P[ok] - z['P'][ok]).max() <= 1e-13 * np.nanmax(z['P'])
    assert abs(total_P - z['total_P']) <= 1e-13 * abs(z['total_P'])
    assert np.array_equal(ux, z['ux']) and np.array_equal(uy, z['uy'])
    assert dux == z['dux'] and duy == z['duy']


def test_reference_flow_on_the_default_grid_golden():
    """lens A on the reference's default 400 x 400 grid: build_nearfield(download=False) ->
    farfield_from_resident_nearfield against what the reference's flow returned (strided sample
    of P, NaN count, arg-max, total_P, axes)"""
    import metalens_amd as ma
    from metalens_amd import _lib
    z = np.load(golden_io.golden_path('farfield_A_lattice.npz'))
    case = np.load(golden_io.golden_path(str(z['nearfield'])))
    lens = golden_io.load_lens(golden_io.golden_path(str(z['lens'])))
    ctx = _lib.default_context()
    out = ma.build_nearfield(float(case['source_x']), float(case['source_y']), float(case['source_z']),
                             str(case['source_pol']), float(case['wavelength']), lens[0], lens[1], lens[2],
                             dipole_moment=float(case['dipole_moment']), c0=float(case['c0']),
                             Z0=float(case['Z0']), ctx=ctx, download=False)
    assert out[0] is None and len(out[4]) == 400
    P, total_P, ux, uy, dux, duy = ma.farfield_from_resident_nearfield(
        out[4], out[5], float(z['wavelength']), out[7], Z0=float(z['Z0']), ctx=ctx)
    # the reference's default grid of this lens has 400 = 16 x 25 samples: its lattice would take sixteen-fold
    # padding to run as the pruned FFT, and `auto` leaves those to the folded GEMMs (measured 1.5 x faster here,
    # profiles/r06_padded_fft_sweep.txt; test_whole_lattice_vs_oracle_flow runs the padded FFT too)
    assert ctx.plan_kernels() == ('folded', 'folded')
    s = int(z['stride'])
    sub = P[3::s, 2::s]
    ok = ~np.isnan(z['P'])
    assert np.array_equal(np.isnan(sub), ~ok)
    assert np.abs(sub[ok] - z['P'][ok]).max() <= 1e-13 * z['P_max']
    assert np.isnan(P).sum() == z['P_nan_count']
    assert abs(total_P - z['total_P']) <= 1e-13 * abs(z['total_P'])
    assert tuple(np.unravel_index(np.nanargmax(P), P.shape)) == tuple(z['P_argmax'])
    assert np.array_equal(ux, z['ux']) and np.array_equal(uy, z['uy'])
    assert dux == z['dux'] and duy == z['duy']


@pytest.mark.parametrize('N,method', [(512, 'auto'), (768, 'auto'), (1024, 'auto'), (384, 'auto'), (960, 'auto'),
                                      (400, 'auto'), (500, 'auto'), (729, 'auto'),
                                      (400, 'fft-streamed'), (500, 'fft-streamed'), (960, 'fft-streamed')])
def test_whole_lattice_vs_oracle_flow(N, method):
    """a lens window of N x N samples: every lattice direction against the oracle's restatement of the
    reference flow (numpy.fft on the host) - ALL N^2 directions, not a sample.  512, 768, 1024:
    multiples of 256 -> both axes run as the pruned FFT with nothing pruned; 384, 960, 400, 500: what
    good_fft_number hands out (nearfield.py:30-36: 2^a 3^b 5^c, not multiples of 256) -> the same FFT on
    the 2 / 4 / 16 / 64 times finer lattice that is one, every 2nd / 4th / 16th / 64th bin wanted (500:
    in five interleaved sub-sequences of 6400 samples) - which `auto` takes up to four-fold padding (beyond that
    the folded GEMMs are faster, metalens_hip.h ML_METHOD_AUTO) and `fft-streamed` always; 729 = 3^6 has no factor
    of two to build on and keeps the folded GEMMs"""
    import metalens_amd as ma
    from metalens_amd import _lib
    from oracle import farfield_oracle
    from test_gpu_parity import _synthetic_lens
    wl = 580e-9
    lens = _synthetic_lens(60e-6, 0.4, wl, switch_deg=9.0)
    x = (np.arange(N) - (N - 1) / 2) * (wl / 2.2)
    args = dict(source_x=0.2e-6, source_y=-0.1e-6, source_z=-lens['source_distance'], source_pol='y',
                wavelength=wl, lens_periphery_summary=lens['lens_periphery_summary'],
                lens_center_summary=lens['lens_center_summary'], hexgridset=lens['hexgridset'],
                x_pts=x, y_pts=x)
    ctx = _lib.default_context()
    fields = ma.build_nearfield(ctx=ctx, **args)          # host copies for the oracle flow
    ma.build_nearfield(ctx=ctx, download=False, **args)   # and the resident set
    ctx.set_method(method)
    try:
        P, total_P, ux, uy, dux, duy = ma.farfield_from_resident_nearfield(x, x, wl, fields[7], ctx=ctx)
    finally:
        ctx.set_method('auto')
    padding = 256 // math.gcd(N, 256)
    assert ctx.plan_kernels() == (('fft', 'fft') if padding <= (4 if method == 'auto' else 64) else ('folded', 'folded'))
    ffts = [np.fft.fft2(np.fft.fftshift(F)) for F in fields[:4]]
    want = farfield_oracle.farfield_from_nearfield(*ffts, x, x, wl, fields[7])
    assert np.array_equal(np.isnan(P), np.isnan(want[0]))
    ok = ~np.isnan(P)
    assert np.abs(P[ok] - want[0][ok]).max() <= 1e-12 * np.nanmax(want[0])
    assert abs(total_P - want[1]) <= 1e-12 * abs(want[1])
    assert np.array_equal(ux, want[2]) and np.array_equal(uy, want[3])
    assert dux == want[4] and duy == want[5]


def test_shape_mismatch_is_refused():
    import metalens_amd as ma
    from metalens_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(0)
    _upload(ctx, [rng.standard_normal((16, 24)) + 0j for _ in range(4)])
    x = np.arange(20) * 2e-7
    with pytest.raises(ValueError):
        ma.farfield_from_resident_nearfield(x, x, 580e-9, 1.459, ctx=ctx)


def test_resident_flow_leaves_the_context_as_it_found_it():
    """the resident flow on a context that a sweep and an fp32 HotPath also use - the
    sweep's per-source sums stay what they were (total_P is taken in a slot of its own), the
    context's precision is put back, and a synthesis that a HotPath left writing modulated fields is
    switched back to plain ones by build_nearfield"""
    import metalens_amd as ma
    from metalens_amd import _lib
    case = np.load(golden_io.golden_path('nearfield_B_straddle_offaxis_y.npz'))
    lens = golden_io.load_lens(golden_io.golden_path(str(case['lens'])))
    wl = float(case['wavelength'])
    x, y = case['x_pts'], case['y_pts']
    u = np.linspace(-0.2, 0.2, 24)
    ctx = _lib.default_context()
    f = abs(float(case['source_z']))
    sources = [(0.0, 0.0, -f, 'x'), (0.0, 0.0, -f, 'y'),
               (float(case['source_x']), float(case['source_y']), -f, 'x')]   # (the fixture's own off-axis source)
    sw = ma.SourceSweep(wl, lens[0], lens[1], lens[2], x, y, u, u, ctx=ctx)
    before = sw.run(sources, cone=0.05)
    # an fp32 hot path with the fused input modulation uses the context next
    hp = ma.HotPath(sources[0], wl, lens[0], lens[1], lens[2], x, y, u, u, ctx=ctx, precision='f32',
                    fuse_modulation=True)
    hp.step()
    hp.sync()
    assert ctx.precision == 'f32'
    out = ma.build_nearfield(0.0, 0.0, -f, 'x', wl, lens[0], lens[1], lens[2], x_pts=x, y_pts=y,
                             ctx=ctx, download=False)
    P, total_P, *_ = ma.farfield_from_resident_nearfield(x, y, wl, out[7], ctx=ctx)
    assert ctx.precision == 'f32'                         # put back
    assert np.isfinite(total_P) and total_P > 0
    # the same flow from a clean context state gives the same map: the fields were plain
    ctx.set_precision('f64')
    out2 = ma.build_nearfield(0.0, 0.0, -f, 'x', wl, lens[0], lens[1], lens[2], x_pts=x, y_pts=y,
                              ctx=ctx, download=False)
    P2, total_P2, *_ = ma.farfield_from_resident_nearfield(x, y, wl, out2[7], ctx=ctx)
    assert np.array_equal(P, P2, equal_nan=True) and total_P == total_P2
    # the sweep's per-source sums are untouched (its P_sum belongs to its own direction grid and plan)
    total = np.zeros(len(sources))
    cone = np.zeros(len(sources))
    _lib.check(ctx.lib.ml_farfield_sums(ctx.handle, None, _lib.dptr(total), _lib.dptr(cone), len(sources)))
    assert np.array_equal(total, before['total_P']) and np.array_equal(cone, before['cone_P'])
