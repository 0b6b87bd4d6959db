"""The whole hot path, GPU-resident: near-field synthesis -> aperture->direction
transform -> (all-reduce) -> projection, with nothing crossing PCIe in between.

This is the composition README.md:27 of the reference describes
(``build_nearfield`` -> ``fft2(fftshift)`` -> ``farfield_from_nearfield``) with the
FFT replaced by the direct transform to a chosen direction grid.  ``bench.py``
times ``step()``; the drop-in functions in nearfield.py / nearfield_farfield.py
are the same kernels with host arrays at the boundary.
"""
import numpy as np

from . import _lib, constants, dist, packing, ties
from .constants import nm
from .grating import n_glass as tabulated_n_glass
from .nearfield import _check_axis, _raise_violation, nearfield_params


class HotPath:
    def __init__(self, source, wavelength, lens_periphery_summary, lens_center_summary,
                 hexgridset, x_pts, y_pts, ux, uy, pair_list=False, dipole_moment=1e-30,
                 c0=None, Z0=None, ctx=None, rank=0, world=1, precision=None,
                 reduce='amplitudes', fuse_modulation=True, method=None):
        """``reduce`` (multi-GPU only): 'amplitudes' all-reduces the two projected complex
        amplitudes (half the payload; the radiation vectors in ``results()`` are then this
        rank's partial sums), 'vectors' all-reduces Nx, Ny, Lx, Ly and projects afterwards."""
        assert reduce in ('amplitudes', 'vectors')
        self.reduce = reduce
        # the plan's stage-1 input modulation rides in the synthesis kernel (metalens_hip.h,
        # ml_nearfield_premodulate); host downloads of the fields are un-modulated first
        self.fuse_modulation = bool(fuse_modulation)
        self.ctx = ctx or _lib.default_context()
        if precision is not None:   # 'f64' | 'f32': arithmetic of the far-field GEMMs
            self.ctx.set_precision(precision)
        if method is not None:      # 'auto' | 'gemm': see _lib.Context.set_method
            self.ctx.set_method(method)
        self.rank, self.world = rank, world
        self.c0 = constants.c0 if c0 is None else c0
        self.Z0 = constants.Z0 if Z0 is None else Z0
        source_x, source_y, source_z, source_pol = source
        assert source_z < 0 and source_pol in ('x', 'y', 'z')
        _check_axis(x_pts, wavelength)
        _check_axis(y_pts, wavelength)
        S = lens_periphery_summary
        wl_nm = int(round(wavelength / nm))
        n_glass = S['gratingcollection_list'][0].grating_list[0].n_glass
        if n_glass == 0:
            n_glass = tabulated_n_glass(wl_nm)
        self.n_glass, self.wavelength = n_glass, wavelength
        packing.upload_tables(self.ctx, S['gratingcollection_list'], hexgridset, wl_nm)
        packing.upload_layout(self.ctx, S, lens_center_summary)
        self._cells = lens_center_summary
        self.dipole_moment = dipole_moment
        self.params = nearfield_params(source_x, source_y, source_z, source_pol, wavelength,
                                       n_glass, dipole_moment, self.c0, self.Z0)
        self.x_all = _lib.f64(x_pts)
        self.y = _lib.f64(y_pts)
        # rows of this rank: mirrored row pairs when the aperture has an even number of rows
        # (both far-field stages fold), one contiguous block otherwise
        self.mirrored = world > 1 and self.x_all.size % 2 == 0
        if self.mirrored:
            # rows through the centre disc cost more near-field time (nearest-cell search, more
            # scattered table gathers); the GEMM cost per row is uniform.  Measured per-row totals
            # (tools/shard_probe.py, 8-way shards): rim 8.0e-4 ms, centre 8.7e-4 ms -> +13 % for
            # a row that lies entirely inside the centre disc
            r_c = float(S['r_min_list'][0])
            xs = self.x_all[:self.x_all.size // 2]
            chord = 2 * np.sqrt(np.maximum(r_c ** 2 - xs ** 2, 0.0))
            span = float(self.y[-1] - self.y[0]) or 1.0
            weights = 1.0 + 0.13 * np.minimum(chord / span, 1.0)
            self.row0, self.row1 = dist.mirrored_block(self.x_all.size, world, rank,
                                                       weights=weights)
            self.rows = dist.mirrored_rows(self.x_all.size, self.row0, self.row1)
        else:
            self.row0, self.row1 = dist.row_block(self.x_all.size, world, rank)
            self.rows = np.arange(self.row0, self.row1)
        self.x_local = np.ascontiguousarray(self.x_all[self.rows])
        self.ux, self.uy = _lib.f64(np.ravel(ux)), _lib.f64(np.ravel(uy))
        self.pair_list = bool(pair_list)
        self.shape = (self.ux.size,) if pair_list else (self.ux.size, self.uy.size)
        self.dxp = x_pts[1] - x_pts[0]
        self.dyp = y_pts[1] - y_pts[0]

    def set_source(self, source):
        """switch to another dipole / plane-wave source; tables, layout and grids stay resident"""
        source_x, source_y, source_z, source_pol = source
        assert source_z < 0 and source_pol in ('x', 'y', 'z')
        if source_z == -float('inf'):
            assert source_pol != 'z'
        self.params = nearfield_params(source_x, source_y, source_z, source_pol, self.wavelength,
                                       self.n_glass, self.dipole_moment, self.c0, self.Z0)

    def step_local(self):
        """near field + transform of this object's rows only (no reduction, no projection)"""
        ctx, lib = self.ctx, self.ctx.lib
        _lib.check(lib.ml_nearfield_premodulate(ctx.handle, int(self.fuse_modulation)))
        _lib.check(lib.ml_farfield_plan(ctx.handle, self.x_all.size, self.y.size, self.dxp,
                                        self.dyp, self.wavelength, self.n_glass,
                                        _lib.dptr(self.ux), self.ux.size, _lib.dptr(self.uy),
                                        self.uy.size, int(self.pair_list)))
        if self.x_local.size:
            _lib.check(lib.ml_nearfield_async(ctx.handle, _lib.byref(self.params),
                                              _lib.dptr(self.x_local), self.x_local.size,
                                              _lib.dptr(self.y), self.y.size))
            if self.mirrored:
                _lib.check(lib.ml_farfield_transform_mirrored_async(ctx.handle, self.row0, 0))
            else:
                _lib.check(lib.ml_farfield_transform_async(ctx.handle, self.row0, 0))

    def step(self):
        """queue one pass of the hot path on the context's stream (asynchronous)"""
        ctx, lib = self.ctx, self.ctx.lib
        _lib.check(lib.ml_nearfield_premodulate(ctx.handle, int(self.fuse_modulation)))
        _lib.check(lib.ml_farfield_plan(ctx.handle, self.x_all.size, self.y.size, self.dxp,
                                        self.dyp, self.wavelength, self.n_glass,
                                        _lib.dptr(self.ux), self.ux.size, _lib.dptr(self.uy),
                                        self.uy.size, int(self.pair_list)))
        if self.x_local.size:
            _lib.check(lib.ml_nearfield_async(ctx.handle, _lib.byref(self.params),
                                              _lib.dptr(self.x_local), self.x_local.size,
                                              _lib.dptr(self.y), self.y.size))
            if self.mirrored:
                _lib.check(lib.ml_farfield_transform_mirrored_async(ctx.handle, self.row0, 0))
            else:
                _lib.check(lib.ml_farfield_transform_async(ctx.handle, self.row0, 0))
        if (self.world > 1 or dist.force_rccl()) and self.reduce == 'amplitudes':
            _lib.check(lib.ml_farfield_project_reduce(ctx.handle, self.Z0))
        else:
            if self.world > 1 or dist.force_rccl():
                _lib.check(lib.ml_farfield_allreduce(ctx.handle))
            _lib.check(lib.ml_farfield_project_async(ctx.handle, self.Z0))

    def sync(self):
        self.ctx.sync()

    def settle_ties(self):
        """True if the last synthesis met samples exactly equidistant from two centre cells and
        the reference's answers (cKDTree, metalens_amd/ties.py) have just been handed to the
        kernels: the pass has to be run once more.  A property of grid and cells, not of the
        source - a sweep pays it once."""
        if not self.x_local.size:
            return False
        return ties.settle(self.ctx, self._cells, self.x_local, self.y) is not None

    def results(self):
        """fetch what the last step left on the GPU; raises the reference's ValueError if a
        sample fell outside the characterisation tables"""
        redo = 1.0 if self.settle_ties() else 0.0
        if self.world > 1 or dist.force_rccl():   # every rank repeats the pass or none does
            redo = float(dist.allreduce_host(self.ctx, [redo], 'max')[0])
        if redo:
            self.step()
            self.sync()
        return self._fetch()

    def _fetch(self):
        ctx, lib = self.ctx, self.ctx.lib
        power = _lib.c_double(0)
        viol = (_lib.BoundViolation * 8)()
        n_viol = _lib.c_int(0)
        _lib.check(lib.ml_nearfield_result(ctx.handle, _lib.byref(power), viol, 8,
                                           _lib.byref(n_viol)))
        if n_viol.value:
            _raise_violation(viol[0], ctx)
        P = np.empty(self.shape)
        a_theta = np.empty(self.shape, dtype=np.complex128)
        a_phi = np.empty(self.shape, dtype=np.complex128)
        _lib.check(lib.ml_farfield_project(ctx.handle, self.Z0, _lib.dptr(P), _lib.dptr(a_theta),
                                           _lib.dptr(a_phi)))
        vec = [np.empty(self.shape, dtype=np.complex128) for _ in range(4)]
        _lib.check(lib.ml_farfield_download(ctx.handle, *[_lib.dptr(v) for v in vec]))
        local_power = power.value * self.dxp * self.dyp
        return {'P': P, 'a_theta': a_theta, 'a_phi': a_phi, 'Nx': vec[0], 'Ny': vec[1],
                'Lx': vec[2], 'Ly': vec[3], 'power_local_rows': local_power}


class HotPath2Stream:
    """The same pass with the aperture rows of this rank split over TWO contexts (= two HIP
    streams) on one GPU: while one half is in the matrix-core-bound GEMMs the other half's
    near-field synthesis (latency / issue-bound, matrix cores idle) runs beside it.  The halves
    are mirrored row-pair shards, exactly as two ranks would own them; the second half's
    radiation vectors are added into the first's on the device (``ml_farfield_add_vectors``,
    ordered by events) before the projection.  Single-GPU use only (world == 1)."""

    def __init__(self, source, wavelength, lens_periphery_summary, lens_center_summary,
                 hexgridset, x_pts, y_pts, ux, uy, ctx=None, precision=None, **kw):
        self.ctx = ctx or _lib.default_context()
        self.ctx_b = _lib.Context(self.ctx.device)
        common = dict(kw)
        self.a = HotPath(source, wavelength, lens_periphery_summary, lens_center_summary,
                         hexgridset, x_pts, y_pts, ux, uy, ctx=self.ctx, rank=0, world=2,
                         precision=precision, **common)
        self.b = HotPath(source, wavelength, lens_periphery_summary, lens_center_summary,
                         hexgridset, x_pts, y_pts, ux, uy, ctx=self.ctx_b, rank=1, world=2,
                         precision=precision, **common)
        self.Z0 = self.a.Z0
        self.n_glass = self.a.n_glass
        self.x_local = self.a.x_all          # all rows are resident on this GPU
        self.shape = self.a.shape

    def set_source(self, source):
        self.a.set_source(source)
        self.b.set_source(source)

    def step(self):
        lib = self.ctx.lib
        self.a.step_local()
        self.b.step_local()
        _lib.check(lib.ml_farfield_add_vectors(self.ctx.handle, self.ctx_b.handle))
        _lib.check(lib.ml_farfield_project_async(self.ctx.handle, self.Z0))

    def sync(self):
        self.ctx_b.sync()
        self.ctx.sync()

    def results(self):
        tied_a, tied_b = self.a.settle_ties(), self.b.settle_ties()
        if tied_a or tied_b:
            self.step()
            self.sync()
        out = self.a._fetch()
        power_b = _lib.c_double(0)
        viol = (_lib.BoundViolation * 8)()
        n_viol = _lib.c_int(0)
        _lib.check(self.ctx_b.lib.ml_nearfield_result(self.ctx_b.handle, _lib.byref(power_b), viol,
                                                      8, _lib.byref(n_viol)))
        if n_viol.value:
            _raise_violation(viol[0], self.ctx_b)
        out['power_local_rows'] += power_b.value * self.a.dxp * self.a.dyp
        return out

    def close(self):
        self.ctx_b.close()
