"""Near-field to far-field transform on MI355X.

``farfield_from_nearfield`` is a drop-in for the reference function of the same
name (reference nearfield_farfield.py:14-75): it takes the caller's
``fft2(fftshift(F))`` arrays and returns ``(P, total_P, ux, uy, dux, duy)``.  The
per-bin projection (``farfield_from_nearfield_helper``, nearfield_farfield.py:77-191)
runs in the HIP kernel ``project_kernel``; the axis bookkeeping, ``fftshift`` and the
finite-sum of :68-74 stay in NumPy exactly as the reference has them.

``FarfieldTransform(..., precision='f32')`` runs the GEMMs on the fp32 matrix cores
(BASELINE.json tolerance 1e-4 instead of 1e-12); storage and everything else stay fp64.

``FarfieldTransform`` / ``farfield_direct`` are new: they evaluate the
aperture -> direction sum that the reference derives in its docstring
(nearfield_farfield.py:97-138) for an ARBITRARY grid (or list) of direction
cosines, as dense complex GEMMs on the fp64 matrix cores, directly from the
GPU-resident near field - no FFT, no host round trip, and linear in the aperture
so that the aperture rows can be sharded over GPUs and summed with one RCCL
all-reduce.  On FFT-lattice directions the result equals the reference's
``fft bin x dx' dy'`` (:135-138).  There is no CPU path.
"""
import numpy as np

from . import _lib, constants


def _check_axis(pts, wavelength):
    d = np.diff(np.asarray(pts, dtype=float))
    assert d.size >= 1
    assert 0 < d[0] < wavelength / 2
    assert d.max() - d.min() <= 1e-9 * np.abs(d).max()


def fft_direction_cosines(n, step, wavelength, n_glass):
    """Direction cosines (in glass) of the n un-shifted FFT bins, aliased into
    the half-range (nearfield_farfield.py:35-39)."""
    u = np.arange(n) * (wavelength / n_glass) / (step * n)
    u[u > u.max() / 2] -= (wavelength / n_glass) / step
    return u


def farfield_from_nearfield(fftEx, fftEy, fftHx, fftHy, xp_list, yp_list, wavelength, n_glass,
                            *, units=None, Z0=None, ctx=None):
    """``fftEx`` is ``fft2(fftshift(Ex))`` and likewise for the others; ``xp_list``,
    ``yp_list`` are the aperture coordinates.  Returns
    ``(P_here_times_r2_over_uz, total_P, ux, uy, dux, duy)`` with the arrays
    fft-shifted, as the reference does.  ``units``: the caller's unit system (the reference takes
    ``nu.Z0``, nearfield_farfield.py:183; ``constants.as_units``), default SI; ``Z0`` overrides it."""
    Z0 = constants.as_units(units).Z0 if Z0 is None else Z0
    dxp = xp_list[1] - xp_list[0]
    dyp = yp_list[1] - yp_list[0]
    num_x, num_y = len(xp_list), len(yp_list)
    assert fftEx.shape == fftEy.shape == fftHx.shape == fftHy.shape == (num_x, num_y)
    _check_axis(xp_list, wavelength)
    _check_axis(yp_list, wavelength)
    ux_list = fft_direction_cosines(num_x, dxp, wavelength, n_glass)
    uy_list = fft_direction_cosines(num_y, dyp, wavelength, n_glass)

    ctx = ctx or _lib.default_context()
    P = np.empty((num_x, num_y), dtype=np.float64)
    ins = [_lib.c128(a) for a in (fftEx, fftEy, fftHx, fftHy)]
    _lib.check(ctx.lib.ml_farfield_lattice_power(
        ctx.handle, num_x, num_y, *[_lib.dptr(a) for a in ins], _lib.dptr(ux_list),
        _lib.dptr(uy_list), dxp, dyp, wavelength, n_glass, Z0, _lib.dptr(P)))

    P = np.fft.fftshift(P)
    ux_list = np.fft.fftshift(ux_list)
    uy_list = np.fft.fftshift(uy_list)
    dux = ux_list[1] - ux_list[0]
    duy = uy_list[1] - uy_list[0]
    ux, uy = np.meshgrid(ux_list, uy_list, indexing='ij', sparse=True)
    total_P = (P * dux * duy)[np.isfinite(P)].sum()
    return P, total_P, ux, uy, dux, duy


def farfield_from_resident_nearfield(xp_list, yp_list, wavelength, n_glass, *, units=None, Z0=None, ctx=None):
    """The reference's whole far-field flow (README.md:27) from the near field that
    ``build_nearfield(..., download=False)`` left on the GPU, without the host round trip:

        Ex, Ey, Hx, Hy, x, y, power, n_glass = reference.build_nearfield(...)
        P, total_P, ux, uy, dux, duy = reference.farfield_from_nearfield(
            fft2(fftshift(Ex)), fft2(fftshift(Ey)), fft2(fftshift(Hx)), fft2(fftshift(Hy)),
            x, y, wavelength, n_glass)

    becomes

        _, _, _, _, x, y, power, n_glass = build_nearfield(..., download=False)
        P, total_P, ux, uy, dux, duy = farfield_from_resident_nearfield(x, y, wavelength, n_glass)

    Same return tuple as the reference (nearfield_farfield.py:75): ``P`` on the WHOLE FFT lattice
    of the aperture, fft-shifted, NaN outside the unit circle (:153-155), ``total_P`` the sum of the
    finite ``P * dux * duy`` (:74), ``ux [N,1]``, ``uy [1,N']`` the shifted direction cosines
    (:35-39,68-70).  The four ``fft2(fftshift(F))`` (:18-20) run on the GPU as the aperture ->
    direction transform over all lattice bins (the pruned FFT of csrc/zfft.hip with nothing pruned
    when the axis length is a multiple of 256 up to 8192, the folded GEMMs otherwise), the
    projection is the same kernel the drop-in ``farfield_from_nearfield`` uses, the finite sum is
    taken on the GPU (fixed-order tree); only ``P`` crosses PCIe.  ``units`` / ``Z0`` as for
    ``farfield_from_nearfield``."""
    Z0 = constants.as_units(units).Z0 if Z0 is None else Z0
    ctx = ctx or _lib.default_context()
    dxp = xp_list[1] - xp_list[0]
    dyp = yp_list[1] - yp_list[0]
    num_x, num_y = len(xp_list), len(yp_list)
    _check_axis(xp_list, wavelength)
    _check_axis(yp_list, wavelength)
    nx, ny = _lib.c_int(0), _lib.c_int(0)
    _lib.check(ctx.lib.ml_fields_shape(ctx.handle, _lib.byref(nx), _lib.byref(ny)))
    if (nx.value, ny.value) != (num_x, num_y):
        raise ValueError('the resident near field is %d x %d but the axes given have %d and %d '
                         'points' % (nx.value, ny.value, num_x, num_y))
    # the reference's lattice, in the reference's own expressions, shifted as it returns it
    ux_list = np.fft.fftshift(fft_direction_cosines(num_x, dxp, wavelength, n_glass))
    uy_list = np.fft.fftshift(fft_direction_cosines(num_y, dyp, wavelength, n_glass))
    # this flow documents fp64; whatever arithmetic the context's other users had chosen is put
    # back afterwards (the plan itself is per call: the next user plans again anyway)
    before = getattr(ctx, 'precision', 'f64')
    t = FarfieldTransform(num_x, num_y, dxp, dyp, wavelength, n_glass, ux_list, uy_list, ctx=ctx,
                          precision='f64')
    lib = ctx.lib
    _lib.check(lib.ml_farfield_transform_async(ctx.handle, 0, 0))
    _lib.check(lib.ml_farfield_project_async(ctx.handle, Z0))
    P = _lib.pinned.empty((num_x, num_y), np.float64)
    _lib.check(lib.ml_farfield_project(ctx.handle, Z0, _lib.dptr(P), None, None))
    # total_P = sum of the finite P * dux * duy, in a slot of its own: a sweep running on the same
    # context keeps its P_sum and its per-source sums
    total = np.zeros(1)
    _lib.check(lib.ml_farfield_total_power(ctx.handle, _lib.dptr(total)))
    dux = ux_list[1] - ux_list[0]
    duy = uy_list[1] - uy_list[0]
    ux, uy = np.meshgrid(ux_list, uy_list, indexing='ij', sparse=True)
    del t
    ctx.set_precision(before)
    return P, float(total[0]), ux, uy, dux, duy


class FarfieldTransform:
    """Direct aperture -> direction transform for one aperture geometry and one set
    of directions.

    ``ux``/``uy`` are direction cosines in glass; with ``pair_list=False`` the far
    field is evaluated on the tensor grid ``ux[:,None] x uy[None,:]``, with
    ``pair_list=True`` at the ``len(ux)`` points ``(ux[d], uy[d])``.

    ``num_x_total`` is the number of aperture rows of the WHOLE aperture; the rows
    resident on this GPU are ``[row0, row0 + local rows)`` (sharded use).
    """

    def __init__(self, num_x_total, num_y, dxp, dyp, wavelength, n_glass, ux, uy,
                 pair_list=False, ctx=None, precision=None):
        self.ctx = ctx or _lib.default_context()
        # 'f64' | 'f32': arithmetic of the GEMMs.  A property of the CONTEXT that plans inherit, so
        # it is set on every construction: an earlier HotPath(precision='f32') on the same context
        # must not leak into a transform that documents 1e-12
        self.ctx.set_precision(precision or 'f64')
        self.ux = _lib.f64(np.ravel(ux))
        self.uy = _lib.f64(np.ravel(uy))
        self.pair_list = bool(pair_list)
        self.shape = (self.ux.size,) if pair_list else (self.ux.size, self.uy.size)
        self.wavelength, self.n_glass = wavelength, n_glass
        _lib.check(self.ctx.lib.ml_farfield_plan(
            self.ctx.handle, num_x_total, num_y, dxp, dyp, wavelength, n_glass,
            _lib.dptr(self.ux), self.ux.size, _lib.dptr(self.uy), self.uy.size, int(pair_list)))

    def transform(self, row0=0, accumulate=False, mirrored=False):
        """radiation vectors of the resident field rows (a partial sum if sharded).
        ``mirrored``: the resident rows are the pairs [row0, row0+h) + [N-row0-h, N-row0)
        (see dist.mirrored_rows) instead of the contiguous block [row0, row0+rows)."""
        fn = (self.ctx.lib.ml_farfield_transform_mirrored if mirrored
              else self.ctx.lib.ml_farfield_transform)
        _lib.check(fn(self.ctx.handle, row0, int(accumulate)))

    def allreduce(self):
        _lib.check(self.ctx.lib.ml_farfield_allreduce(self.ctx.handle))

    def radiation_vectors(self):
        out = [np.empty(self.shape, dtype=np.complex128) for _ in range(4)]
        _lib.check(self.ctx.lib.ml_farfield_download(self.ctx.handle, *[_lib.dptr(a) for a in out]))
        return dict(zip(('Nx', 'Ny', 'Lx', 'Ly'), out))

    def project(self, Z0=None, units=None):
        """-> P (= power x r^2 / uz per unit dux duy, NaN outside the unit circle) and the
        two complex far-field amplitudes ``L_phi + Z N_theta`` (prop. to E_theta) and
        ``L_theta - Z N_phi`` (prop. to -E_phi) of nearfield_farfield.py:184-185."""
        Z0 = constants.as_units(units).Z0 if Z0 is None else Z0
        P = np.empty(self.shape, dtype=np.float64)
        a_theta = np.empty(self.shape, dtype=np.complex128)
        a_phi = np.empty(self.shape, dtype=np.complex128)
        _lib.check(self.ctx.lib.ml_farfield_project(self.ctx.handle, Z0, _lib.dptr(P),
                                                    _lib.dptr(a_theta), _lib.dptr(a_phi)))
        return P, a_theta, a_phi


def farfield_direct(Ex, Ey, Hx, Hy, xp_list, yp_list, wavelength, n_glass, ux, uy,
                    *, pair_list=False, units=None, Z0=None, ctx=None, precision=None):
    """One-shot convenience: far field of host arrays ``Ex..Hy`` (or of the field set
    already resident on the GPU if ``Ex is None``) at the given directions.  Returns a
    dict with ``Nx, Ny, Lx, Ly, P, a_theta, a_phi``.  ``precision``: 'f64' (default, 1e-12)
    or 'f32' (the fp32 matrix-core GEMMs, 1e-4); whatever an earlier user of the context chose
    does not carry over."""
    ctx = ctx or _lib.default_context()
    dxp = xp_list[1] - xp_list[0]
    dyp = yp_list[1] - yp_list[0]
    _check_axis(xp_list, wavelength)
    _check_axis(yp_list, wavelength)
    if Ex is not None:
        arrs = [_lib.c128(a) for a in (Ex, Ey, Hx, Hy)]
        assert arrs[0].shape == arrs[1].shape == arrs[2].shape == arrs[3].shape == (len(xp_list), len(yp_list))
        _lib.check(ctx.lib.ml_fields_upload(ctx.handle, len(xp_list), len(yp_list),
                                            *[_lib.dptr(a) for a in arrs]))
    t = FarfieldTransform(len(xp_list), len(yp_list), dxp, dyp, wavelength, n_glass, ux, uy,
                          pair_list=pair_list, ctx=ctx, precision=precision)
    t.transform()
    out = t.radiation_vectors()
    out['P'], out['a_theta'], out['a_phi'] = t.project(Z0, units)
    return out
