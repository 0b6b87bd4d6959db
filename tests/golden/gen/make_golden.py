#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (it does not exist on the GPU box and nothing else in this repo
reads it).  The reference's Python files are imported unmodified; four
third-party packages they import are absent from the image, so stand-ins from
tests/golden/gen/standins/ are put on sys.path first:

* numericalunits - constants only, plain SI; c0 and Z0 enter the arithmetic and
  are recorded in every fixture;
* dxfwrite, ezdxf, svgwrite - imported at module level by design_collimator.py,
  never executed on the near-field / far-field path.

S4 is unavailable, so characterisation tables are synthetic
(metalens_amd/synthetic.py writes the records into the reference's own
Grating / GratingCollection / HexGridSet objects).

What is committed: the *.npz outputs (data) and this script.  No reference
source or bytecode is copied.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen/make_golden.py
"""
import contextlib
import io
import math
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..', '..'))
REFERENCE = os.environ.get('METALENS_REFERENCE', '/root/reference')
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, 'standins'), REFERENCE, REPO, os.path.join(REPO, 'tests')]
warnings.filterwarnings('ignore')

import numpy as np  # noqa: E402
import scipy  # noqa: E402
from scipy.interpolate import RegularGridInterpolator  # noqa: E402

import numericalunits as nu  # the stand-in  # noqa: E402
import grating as ref_grating  # noqa: E402
import lens_center as ref_lens_center  # noqa: E402
import design_collimator as ref_design  # noqa: E402
import nearfield as ref_nearfield  # noqa: E402
import nearfield_farfield as ref_farfield  # noqa: E402

from metalens_amd import synthetic  # noqa: E402
import golden_io  # noqa: E402

nm, um = 1e-9, 1e-6
degree = math.pi / 180
inf = float('inf')
OUT = os.path.join(REPO, 'tests', 'golden')
REF_CLASSES = (ref_grating.Grating, ref_grating.GratingCollection, ref_lens_center.HexGridSet)
META = dict(c0=nu.c0, Z0=nu.Z0, numpy=np.__version__, scipy=scipy.__version__)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print('%-34s %8.1f KiB' % (name, os.path.getsize(path) / 1024))


def crop_cells(cells, x_pts, y_pts, margin=2 * um):
    keep = ((cells[:, 0] > x_pts[0] - margin) & (cells[:, 0] < x_pts[-1] + margin)
            & (cells[:, 1] > y_pts[0] - margin) & (cells[:, 1] < y_pts[-1] + margin))
    if keep.sum() < 4:
        # the reference builds its kd-tree even when no sample is in the centre
        keep[:4] = True
    return cells[keep]


def window(cx, cy, nx, ny, wavelength):
    pitch = wavelength / 2.2
    x = cx + (np.arange(nx) - (nx - 1) / 2) * pitch
    y = cy + (np.arange(ny) - (ny - 1) / 2) * pitch
    return x, y


def run_case(name, lens_name, lens, cells, source, x_pts, y_pts, stride=None, extra=None):
    sx, sy, sz, pol = source
    args = dict(source_x=sx, source_y=sy, source_z=sz, source_pol=pol,
                wavelength=lens['wavelength'],
                lens_periphery_summary=lens['lens_periphery_summary'],
                lens_center_summary=cells, hexgridset=lens['hexgridset'],
                x_pts=x_pts, y_pts=y_pts)
    Ex, Ey, Hx, Hy, xo, yo, power, n_glass = quiet(ref_nearfield.build_nearfield, **args)
    out = dict(lens=lens_name, source_x=sx, source_y=sy, source_z=sz, source_pol=pol,
               wavelength=lens['wavelength'], dipole_moment=1e-30, x_pts=xo, y_pts=yo,
               power=power, n_glass=n_glass, **META)
    if stride is None:
        out.update(Ex=Ex, Ey=Ey, Hx=Hx, Hy=Hy)
    else:
        sl = (slice(stride // 2, None, stride), slice(stride // 3, None, stride))
        out.update(stride=stride, Ex=Ex[sl], Ey=Ey[sl], Hx=Hx[sl], Hy=Hy[sl],
                   sums=np.array([Ex.sum(), Ey.sum(), Hx.sum(), Hy.sum()]),
                   norms=np.array([np.abs(F).max() for F in (Ex, Ey, Hx, Hy)]),
                   nonzero=np.array([(F != 0).sum() for F in (Ex, Ey, Hx, Hy)]))
    if extra:
        out.update(extra)
    save(name, **out)
    return Ex, Ey, Hx, Hy, xo, yo, power, n_glass


def main():
    os.makedirs(OUT, exist_ok=True)
    wl = 580 * nm

    # ------------------------------------------------------------------ lens A
    # config-1-like: 100 um diameter, NA 0.3, reference default grid (400 x 400)
    lensA = synthetic.make_lens(REF_CLASSES, ref_design.make_design, radius=50 * um,
                                numerical_aperture=0.3, wavelength=wl,
                                switch_angle=8 * degree, num_gratings=14, num_entries=10)
    save('lensA.npz', **golden_io.pack_lens(lensA['lens_periphery_summary'],
                                            lensA['lens_center_summary'], lensA['hexgridset']),
         source_distance=lensA['source_distance'], r_for_switch=lensA['r_for_switch'], **META)
    fA = lensA['source_distance']
    Ex, Ey, Hx, Hy, xA, yA, pA, nA = run_case(
        'nearfield_A_default_grid.npz', 'lensA.npz', lensA, lensA['lens_center_summary'],
        (0.0, 0.0, -fA, 'x'), None, None, stride=7)
    assert len(xA) == 400
    # build_nearfield_big must reproduce build_nearfield (strips are independent)
    big = quiet(ref_nearfield.build_nearfield_big, source_x=0.0, source_y=0.0, source_z=-fA,
                source_pol='x', wavelength=wl,
                lens_periphery_summary=lensA['lens_periphery_summary'],
                lens_center_summary=lensA['lens_center_summary'],
                hexgridset=lensA['hexgridset'], x_pts=xA, y_pts=yA)
    save('nearfield_A_big_vs_single.npz',
         max_abs_diff=np.array([np.abs(a - b).max() for a, b in zip(big[:4], (Ex, Ey, Hx, Hy))]),
         power_big=big[6], power_single=pA, **META)

    # far field of lens A on the FFT lattice, strided + reductions
    fEx, fEy, fHx, fHy = (np.fft.fft2(np.fft.fftshift(F)) for F in (Ex, Ey, Hx, Hy))
    P, total_P, ux, uy, dux, duy = quiet(ref_farfield.farfield_from_nearfield,
                                         fEx, fEy, fHx, fHy, xA, yA, wl, nA)
    sl = (slice(3, None, 7), slice(2, None, 7))
    save('farfield_A_lattice.npz', lens='lensA.npz', nearfield='nearfield_A_default_grid.npz',
         P=P[sl], stride=7, total_P=total_P, ux=ux, uy=uy, dux=dux, duy=duy,
         P_nansum=np.nansum(P), P_nan_count=np.isnan(P).sum(), P_max=np.nanmax(P),
         P_argmax=np.array(np.unravel_index(np.nanargmax(P), P.shape)),
         wavelength=wl, n_glass=nA, **META)

    # ------------------------------------------------------------------ lens B
    # 0.5 mm diameter, NA 0.5; small windows at chosen places
    lensB = synthetic.make_lens(REF_CLASSES, ref_design.make_design, radius=250 * um,
                                numerical_aperture=0.5, wavelength=wl,
                                switch_angle=12 * degree, num_gratings=16, num_entries=12)
    fB = lensB['source_distance']
    rsw = lensB['r_for_switch']
    windows = {
        'center': window(10 * um, -5 * um, 48, 40, wl),
        'straddle': window(rsw * math.cos(0.5), rsw * math.sin(0.5), 48, 40, wl),
        'periphery': window(200 * um * math.cos(1.75), 200 * um * math.sin(1.75), 40, 48, wl),
        'edge': window(251 * um * math.cos(-0.8), 251 * um * math.sin(-0.8), 48, 40, wl),
    }
    all_x = np.hstack([w[0] for w in windows.values()])
    all_y = np.hstack([w[1] for w in windows.values()])
    # one cropped cell list serving all windows: cells within 2 um of any window
    cells = lensB['lens_center_summary']
    keep = np.zeros(len(cells), dtype=bool)
    for wx, wy in windows.values():
        c = crop_cells(cells, wx, wy)
        keep |= np.isin(cells[:, 0] + 1j * cells[:, 1], c[:, 0] + 1j * c[:, 1])
    cellsB = cells[keep]
    save('lensB.npz', **golden_io.pack_lens(lensB['lens_periphery_summary'], cellsB,
                                            lensB['hexgridset']),
         source_distance=fB, r_for_switch=rsw, full_cell_count=len(cells), **META)
    sources = {
        'onaxis_x': (0.0, 0.0, -fB, 'x'),
        'offaxis_y': (3 * um, -2 * um, -fB * 1.02, 'y'),
        'offaxis_z': (-2 * um, 4 * um, -fB, 'z'),
        'plane_x': (0.0, 0.0, -inf, 'x'),
        'plane_y': (0.0, 0.0, -inf, 'y'),
    }
    plan = [('center', 'onaxis_x'), ('center', 'offaxis_z'), ('center', 'plane_y'),
            ('straddle', 'onaxis_x'), ('straddle', 'offaxis_y'), ('straddle', 'plane_x'),
            ('periphery', 'onaxis_x'), ('periphery', 'offaxis_y'), ('periphery', 'offaxis_z'),
            ('straddle', 'plane_y'),
            ('edge', 'onaxis_x'), ('edge', 'offaxis_y')]
    results = {}
    for wname, sname in plan:
        wx, wy = windows[wname]
        results[(wname, sname)] = run_case('nearfield_B_%s_%s.npz' % (wname, sname), 'lensB.npz',
                                           lensB, cellsB, sources[sname], wx, wy)

    # ------------------------------------------ far field of one window (full)
    Ex, Ey, Hx, Hy, wx, wy, _, nB = results[('periphery', 'onaxis_x')]
    ffts = [np.fft.fft2(np.fft.fftshift(F)) for F in (Ex, Ey, Hx, Hy)]
    P, total_P, ux, uy, dux, duy = quiet(ref_farfield.farfield_from_nearfield,
                                         *ffts, wx, wy, wl, nB)
    dA = (wx[1] - wx[0]) * (wy[1] - wy[0])
    save('farfield_B_periphery_window.npz', nearfield='nearfield_B_periphery_onaxis_x.npz',
         fftEx=ffts[0], fftEy=ffts[1], fftHx=ffts[2], fftHy=ffts[3],
         P=P, total_P=total_P, ux=ux, uy=uy, dux=dux, duy=duy, wavelength=wl, n_glass=nB,
         # known answers for the direct aperture->direction sum (SURVEY §8c iv):
         # N, L = FFT bin x dA with the reference's signs (nearfield_farfield.py:135-138)
         Nx=-ffts[3] * dA, Ny=ffts[2] * dA, Lx=ffts[1] * dA, Ly=-ffts[0] * dA, **META)

    # --------------------------------------------------------- negative tests
    neg = {}

    def expect(label, exc_type, **override):
        wx, wy = windows['periphery']
        a = dict(source_x=0.0, source_y=0.0, source_z=-fB, source_pol='x', wavelength=wl,
                 lens_periphery_summary=lensB['lens_periphery_summary'],
                 lens_center_summary=cellsB, hexgridset=lensB['hexgridset'],
                 x_pts=wx, y_pts=wy)
        a.update(override)
        try:
            quiet(ref_nearfield.build_nearfield, **a)
        except exc_type as e:
            neg[label + '_type'] = np.array(type(e).__name__)
            neg[label + '_msg'] = np.array(str(e.args[0]) if e.args else '')
            neg[label + '_vals'] = np.array([float(v) for v in e.args[1:]], dtype=float)
            return
        raise RuntimeError('reference did not raise for ' + label)

    wx, wy = windows['periphery']
    expect('coarse_pitch', AssertionError, x_pts=wx[::2])
    expect('nonuniform', AssertionError, x_pts=np.hstack((wx[:-1], wx[-1] + 1e-9)))
    expect('source_above', AssertionError, source_z=1e-6)
    expect('bad_pol', AssertionError, source_pol='s')
    expect('plane_z', AssertionError, source_z=-inf, source_pol='z')
    # a source far off axis drives u'_x beyond the tabulated range of the collection
    expect('ux_overrun', ValueError, source_x=-180 * um)
    expect('uy_overrun', ValueError, source_y=150 * um, source_x=100 * um)
    # a normally incident plane wave is outside the outer collections' tables
    expect('plane_wave_overrun', ValueError, source_z=-inf)
    wxc, wyc = windows['center']
    expect('center_overrun', ValueError, source_x=-60 * um, source_z=-60 * um, x_pts=wxc, y_pts=wyc)
    try:
        ref_grating.n_glass(532)
    except ValueError as e:
        neg['bad_wavelength_msg'] = np.array(str(e.args[0]))
    save('negative_cases.npz', lens='lensB.npz', **neg, **META)

    # ------------------------------------------------- lenses C and D: other order sets
    # Every fixture above tabulates the orders (0,0), (-1,0), (+1,0).  characterize() records, per
    # direction, every order that propagates in air (grating.lua:417-423): lens C carries exactly
    # those lists (eleven orders in the inner collection, seven in the outer, three in the centre,
    # absent entries zero-filled by build_interpolators); lens D a five-order and a three-order
    # collection ({-3 ... +1} inside, {-2, -1, 0} outside) around a one-order centre.
    for tag, per, cen in (('C', 'physical', 'physical'),
                          ('D', (((-3, 0), (-2, 0), (-1, 0), (0, 0), (1, 0)), ((-2, 0), (-1, 0), (0, 0))),
                           ((0, 0),))):
        lens = synthetic.make_lens(REF_CLASSES, ref_design.make_design, radius=60 * um,
                                   numerical_aperture=0.42, wavelength=wl, switch_angle=10 * degree,
                                   num_gratings=12, num_entries=8, periphery_orders=per,
                                   center_orders=cen)
        S = lens['lens_periphery_summary']
        assert len(S['gratingcollection_list']) == 2
        f = lens['source_distance']
        rsw = lens['r_for_switch']
        gc_of_ring = np.asarray(S['gratingcollection_index_here_list'])
        r_join = float(np.asarray(S['r_min_list'])[int(np.argmax(gc_of_ring == 1))])
        wins = {
            'center': window(-4 * um, 3 * um, 40, 48, wl),
            'straddle': window(rsw * math.cos(2.2), rsw * math.sin(2.2), 48, 40, wl),
            'join': window(r_join * math.cos(-1.1), r_join * math.sin(-1.1), 48, 40, wl),
            'edge': window(60.5 * um * math.cos(0.4), 60.5 * um * math.sin(0.4), 40, 48, wl),
        }
        cells = lens['lens_center_summary']
        keep = np.zeros(len(cells), dtype=bool)
        for wx, wy in wins.values():
            c = crop_cells(cells, wx, wy)
            keep |= np.isin(cells[:, 0] + 1j * cells[:, 1], c[:, 0] + 1j * c[:, 1])
        cells = cells[keep]
        name = 'lens%s.npz' % tag
        save(name, **golden_io.pack_lens(S, cells, lens['hexgridset']), source_distance=f,
             r_for_switch=rsw, r_join=r_join, **META)
        srcs = {'onaxis_x': (0.0, 0.0, -f, 'x'), 'offaxis_z': (2 * um, -3 * um, -f * 0.97, 'z'),
                'offaxis_y': (-1.5 * um, 1 * um, -f, 'y'), 'plane_x': (0.0, 0.0, -inf, 'x')}
        for wname, sname in (('center', 'plane_x'), ('center', 'offaxis_z'), ('straddle', 'onaxis_x'),
                             ('join', 'onaxis_x'), ('join', 'offaxis_y'), ('edge', 'offaxis_z')):
            wx, wy = wins[wname]
            run_case('nearfield_%s_%s_%s.npz' % (tag, wname, sname), name, lens, cells, srcs[sname], wx, wy)

    # ------------------------------------------------- raw scipy RGI samples
    rng = np.random.default_rng(12345)
    ax0 = np.array([-0.3, -0.1, 0.05, 0.2, 0.45])
    ax1 = np.linspace(-0.2, 0.2, 5)
    ax2 = np.hstack((0.99 * 600e-9, np.linspace(600e-9, 900e-9, 7), 1.01 * 900e-9))
    vals = rng.standard_normal((5, 5, 9)) + 1j * rng.standard_normal((5, 5, 9))
    pts = np.column_stack((rng.uniform(ax0[0], ax0[-1], 300), rng.uniform(ax1[0], ax1[-1], 300),
                           rng.uniform(ax2[0], ax2[-1], 300)))
    # exact nodes, upper edges, lower edges
    special = np.array([[ax0[2], ax1[3], ax2[4]], [ax0[-1], ax1[-1], ax2[-1]],
                        [ax0[0], ax1[0], ax2[0]], [ax0[-1], 0.01, ax2[1]],
                        [0.0, ax1[-1], ax2[-2]], [ax0[1], ax1[0], ax2[-1]]])
    pts = np.vstack((pts, special))
    f = RegularGridInterpolator((ax0, ax1, ax2), vals)
    save('rgi_samples.npz', axis0=ax0, axis1=ax1, axis2=ax2, values=vals, points=pts,
         result=f(pts), **META)

    # good_fft_number known answers (reference nearfield.py:30-36)
    goals = np.array([1, 2, 7, 11, 379.3, 400, 401, 1000.5, 4097, 8191.2, 8192, 99999])
    save('good_fft_number.npz', goals=goals,
         answers=np.array([ref_nearfield.good_fft_number(g) for g in goals]))


if __name__ == '__main__':
    main()
