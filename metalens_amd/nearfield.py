"""Near-field of a grating-based metasurface lens, synthesised on MI355X.

Drop-in for the reference's ``nearfield.py``: ``build_nearfield``,
``build_nearfield_big`` and ``good_fft_number`` keep the reference's signatures,
argument meaning, return tuples, assertions and ``ValueError`` messages
(reference nearfield.py:30-36,66-68,84-85,106-109,294-305,412-419,480,482-516).
The arithmetic runs in the hand-written HIP kernels ``nearfield_geometry_kernel``
(once per grid and layout) and ``nearfield_field_kernel`` (csrc/nearfield_fast.hip;
csrc/nearfield.hip is their launch code) through the C ABI of include/metalens_hip.h; this module
only evaluates the scalar set-up with the reference's own expressions, flattens
the inputs (packing.py) and converts the kernel's bound-check report back into
the reference's exceptions.  There is no CPU path.

Extra keyword-only arguments (not in the reference):
``units``       the caller's unit system: an object with ``nm``, ``c0``, ``Z0`` (``C``, ``m``), e.g. the
                ``numericalunits`` module the reference's own callers hold; default SI
                (``metalens_amd.constants``).  Decides the table key ``int(round(wavelength / nm))``
                (nearfield.py:111), the default ``c0`` / ``Z0`` / dipole moment and the nanometres of
                the period bound errors;
``c0``, ``Z0``  physical constants in the caller's units, default those of ``units``;
``ctx``         the ``_lib.Context`` (GPU) to use, default the process-wide one;
``download``    if False the fields stay resident on the GPU for the far-field
                transform and ``None`` is returned in their place.
"""
from math import pi

import numpy as np

from . import _lib, constants, packing, ties
from .grating import n_glass as tabulated_n_glass
from .prepared import PreparedLens

inf = float('inf')

_CHECK_MESSAGES = ('need to calculate at smaller ux!', 'need to calculate at bigger ux!',
                   'need to calculate at smaller uy!', 'need to calculate at bigger uy!',
                   'need to calculate at smaller grating_period!',
                   'need to calculate at bigger grating_period!')


def good_fft_number(goal):
    """Smallest integer >= goal whose only prime factors are 2, 3, 5
    (reference nearfield.py:30-36; same ``goal < 1e5`` precondition)."""
    assert goal < 1e5
    best = None
    p2 = 1
    while best is None or p2 < best:
        p3 = p2
        while best is None or p3 < best:
            p5 = p3
            while p5 < goal:
                p5 *= 5
            if best is None or p5 < best:
                best = p5
            p3 *= 3
        p2 *= 2
    return best


def _check_axis(pts, wavelength):
    """ascending, uniform to 1e-9, finer than half a wavelength (nearfield.py:106-109)"""
    d = np.diff(np.asarray(pts, dtype=float))
    assert d.size >= 1
    assert 0 < d[0] < wavelength / 2
    assert d.max() - d.min() <= 1e-9 * np.abs(d).max()


def nearfield_params(source_x, source_y, source_z, source_pol, wavelength, n_glass,
                     dipole_moment, c0, Z0):
    """The per-call scalars, each written the way the reference writes it so that the
    kernel starts from bit-identical values."""
    p = _lib.NearfieldParams()
    p.source_x, p.source_y, p.source_z = source_x, source_y, source_z
    dz = 0 - source_z                                              # :174
    p.dz = dz
    p.dz2 = dz ** 2                                                # :175
    p.source_z2 = source_z ** 2                                    # :340
    pol = {'x': [1, 0, 0], 'y': [0, 1, 0], 'z': [0, 0, 1]}[source_pol]   # :215
    p.pol[0], p.pol[1], p.pol[2] = pol
    kvac = 2 * pi / wavelength                                     # :115
    k_glass = 2 * pi * n_glass / wavelength                        # :114
    p.kvac, p.kvac2 = kvac, kvac ** 2
    p.k_glass, p.k_glass2 = k_glass, k_glass ** 2
    p.n_glass = n_glass
    p.Z0 = Z0
    p.H_coef = c0 * (2 * pi / wavelength) ** 2 * dipole_moment / (4 * pi)   # :213
    p.dipole_moment = dipole_moment
    p.plane_wave = 1 if source_z == -inf else 0
    return p


def _raise_violation(v, ctx, nm=constants.nm):
    """first violated bound -> the reference's ValueError (nearfield.py:294-305,412-419)"""
    msg = _CHECK_MESSAGES[v.check]
    if v.check >= 4:
        raise ValueError(msg, v.value / nm, v.bound / nm)
    raise ValueError(msg, v.value, v.bound)


def build_nearfield(source_x, source_y, source_z, source_pol, wavelength,
                    lens_periphery_summary, lens_center_summary, hexgridset,
                    x_pts=None, y_pts=None, dipole_moment=None,
                    *, units=None, c0=None, Z0=None, ctx=None, download=True):
    """Ex, Ey, Hx, Hy just past the lens for a dipole at (source_x, source_y,
    source_z<0) polarised along ``source_pol`` in 'x','y','z', or for a normally
    incident plane wave if ``source_z == -inf`` (then ``dipole_moment`` is the
    E-field magnitude).  Returns ``(Ex, Ey, Hx, Hy, x_pts, y_pts,
    power_passing_through_lens, n_glass)`` like the reference (nearfield.py:480)."""
    assert source_z < 0
    assert source_pol in ('x', 'y', 'z')
    units = constants.as_units(units)
    c0 = units.c0 if c0 is None else c0
    Z0 = units.Z0 if Z0 is None else Z0
    if dipole_moment is None:
        dipole_moment = constants.default_dipole_moment(units)   # 1e-30 C m (nearfield.py:68)
    wavelength_in_nm = int(round(wavelength / units.nm))
    # a PreparedLens in place of the periphery summary (prepared.py): hashed and uploaded once
    prepared = lens_periphery_summary if isinstance(lens_periphery_summary, PreparedLens) else None
    if prepared is not None:
        lens_periphery_summary = prepared.lens_periphery_summary
        lens_center_summary, hexgridset = prepared.lens_center_summary, prepared.hexgridset
    S = lens_periphery_summary
    gc_list = S['gratingcollection_list']
    lens_max_r = S['r_max_list'][-1]
    if x_pts is None:
        x_pts = np.linspace(-lens_max_r, lens_max_r,
                            num=good_fft_number(2 * lens_max_r / (wavelength / 2.2)))
    if y_pts is None:
        y_pts = np.linspace(-lens_max_r, lens_max_r,
                            num=good_fft_number(2 * lens_max_r / (wavelength / 2.2)))
    _check_axis(x_pts, wavelength)
    _check_axis(y_pts, wavelength)
    if source_z == -inf:
        assert source_pol != 'z'

    n_glass = gc_list[0].grating_list[0].n_glass
    if n_glass == 0:
        n_glass = tabulated_n_glass(wavelength_in_nm)

    xs, ys = _lib.f64(x_pts), _lib.f64(y_pts)
    # no sample inside the lens: the reference returns zeros and an integer 0 power
    # (nearfield.py:130-134)
    nearest_r = (np.abs(xs).min() ** 2 + np.abs(ys).min() ** 2) ** 0.5
    if nearest_r > lens_max_r:
        zero = np.zeros((xs.size, ys.size), dtype=complex)
        return zero, zero, zero, zero, x_pts, y_pts, 0, n_glass

    if prepared is not None:
        ctx = ctx or prepared.ctx
        prepared.make_resident(ctx, wavelength_in_nm)
    else:
        ctx = ctx or _lib.default_context()
        packing.upload_tables(ctx, gc_list, hexgridset, wavelength_in_nm)
        packing.upload_layout(ctx, S, lens_center_summary)
    # the drop-in hands back (or leaves resident) the PLAIN fields: a HotPath that shared this
    # context may have left the synthesis writing exp(-i pi (i+j)) modulated ones
    _lib.check(ctx.lib.ml_nearfield_premodulate(ctx.handle, 0))
    p = nearfield_params(source_x, source_y, source_z, source_pol, wavelength, n_glass,
                         dipole_moment, c0, Z0)
    power = _lib.c_double(0)
    max_v = 64
    viol = (_lib.BoundViolation * max_v)()
    n_viol = _lib.c_int(0)
    known = None
    for attempt in range(3):
        _lib.check(ctx.lib.ml_nearfield(ctx.handle, _lib.byref(p), _lib.dptr(xs), xs.size,
                                        _lib.dptr(ys), ys.size, _lib.byref(power), viol, max_v,
                                        _lib.byref(n_viol)))
        # samples exactly equidistant from two centre cells: the reference's choice is
        # cKDTree's (nearfield.py:363-364); ask it about those samples and run again
        known = ties.settle(ctx, lens_center_summary, xs, ys, known)
        if known is None:
            break
    else:
        raise _lib.MetalensHipError('nearest-cell ties were still being reported after three '
                                    'passes (%d samples)' % ties.pending(ctx).size)
    if n_viol.value:
        _raise_violation(viol[0], ctx, units.nm)
    power_passing_through_lens = power.value * (x_pts[1] - x_pts[0]) * (y_pts[1] - y_pts[0])
    if not download:
        return None, None, None, None, x_pts, y_pts, power_passing_through_lens, n_glass
    # page-locked storage (recycled through _lib.pinned): the copy runs at the PCIe rate
    out = [_lib.pinned.empty((xs.size, ys.size), np.complex128) for _ in range(4)]
    _lib.check(ctx.lib.ml_fields_download(ctx.handle, *[_lib.dptr(a) for a in out]))
    return out[0], out[1], out[2], out[3], x_pts, y_pts, power_passing_through_lens, n_glass


def build_nearfield_big(source_x, source_y, source_z, source_pol, wavelength,
                        lens_periphery_summary, lens_center_summary, hexgridset,
                        x_pts=None, y_pts=None, dipole_moment=None,
                        *, units=None, c0=None, Z0=None, ctx=None, pts_at_a_time=1e7):
    """Strip driver with the reference's tiling contract (nearfield.py:482-516):
    y-strips of ``int(pts_at_a_time / len(x_pts))`` samples, fields concatenated,
    strip powers added.  Like the reference it needs explicit ``x_pts`` and
    ``y_pts``.  (On a 288 GB MI355X the strips are not needed for memory; they are
    kept because they define the order in which the power is summed.)"""
    strip = int(pts_at_a_time / x_pts.size)
    shape = (x_pts.size, y_pts.size)
    Ex, Ey, Hx, Hy = (np.zeros(shape, dtype=complex) for _ in range(4))
    power_passing_through_lens = 0
    n_glass = None
    start = 0
    while start < y_pts.size:
        end = min(start + strip, y_pts.size)
        ex, ey, hx, hy, _, _, p_now, n_glass = build_nearfield(
            source_x, source_y, source_z, source_pol, wavelength, lens_periphery_summary,
            lens_center_summary, hexgridset, x_pts=x_pts, y_pts=y_pts[start:end],
            dipole_moment=dipole_moment, units=units, c0=c0, Z0=Z0, ctx=ctx)
        Ex[:, start:end] = ex
        Ey[:, start:end] = ey
        Hx[:, start:end] = hx
        Hy[:, start:end] = hy
        power_passing_through_lens += p_now
        start = end
    return Ex, Ey, Hx, Hy, x_pts, y_pts, power_passing_through_lens, n_glass
