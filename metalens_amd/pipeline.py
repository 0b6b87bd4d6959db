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
from .grating import n_glass as tabulated_n_glass
from .nearfield import _check_axis, _raise_violation, nearfield_params


def _check_source(source):
    """the reference's assertions on a source (nearfield.py:84-85,224): below the lens, polarised
    along x, y or z, and no z-polarised plane wave"""
    source_x, source_y, source_z, source_pol = source
    assert source_z < 0 and source_pol in ('x', 'y', 'z')
    if source_z == -float('inf'):
        assert source_pol != 'z'
    return source_x, source_y, source_z, source_pol


class HotPath:
    def __init__(self, source, wavelength, lens_periphery_summary, lens_center_summary,
                 hexgridset, x_pts, y_pts, ux, uy, pair_list=False, dipole_moment=None,
                 c0=None, Z0=None, ctx=None, rank=0, world=1, precision=None,
                 reduce='amplitudes', fuse_modulation=True, method=None, sharding='auto', units=None):
        """``reduce`` (multi-GPU only): 'amplitudes' sums the two projected complex amplitudes over
        the ranks - by a reduce-scatter over blocks of direction rows, each rank taking the power of
        its block (``results()`` gathers the whole map); the radiation vectors in ``results()`` are
        then this rank's partial sums - 'amplitudes-allreduce' does the same by an all-reduce (twice
        the bytes per rank), 'vectors' all-reduces Nx, Ny, Lx, Ly and projects afterwards."""
        assert reduce in ('amplitudes', 'amplitudes-allreduce', 'vectors', 'none')
        allreduce = reduce == 'amplitudes-allreduce'   # (comparison runs: the all-reduce instead of the reduce-scatter)
        reduce = 'amplitudes' if allreduce else reduce
        assert sharding in ('auto', 'interleaved', 'mirrored', 'rows')
        self.reduce = reduce
        # the plan's stage-1 input modulation rides in the synthesis kernel (metalens_hip.h,
        # ml_nearfield_premodulate); host downloads of the fields are un-modulated first
        self.fuse_modulation = bool(fuse_modulation)
        self.ctx = ctx or _lib.default_context()
        # both are properties of the context that later plans inherit: set them on every
        # construction so that an earlier HotPath(precision='f32') cannot leak into this one
        self.ctx.set_precision(precision or 'f64')   # 'f64' | 'f32': arithmetic of the GEMMs
        self.ctx.set_method(method or 'auto')        # 'auto' | 'gemm' | 'fft-streamed': _lib.Context.set_method
        if hasattr(self.ctx.lib, 'ml_comm_set_reduce'):
            _lib.check(self.ctx.lib.ml_comm_set_reduce(self.ctx.handle, int(allreduce)))
        self.rank, self.world = rank, world
        # ``units``: the caller's unit system (nearfield.build_nearfield); default SI
        self.units = constants.as_units(units)
        self.c0 = self.units.c0 if c0 is None else c0
        self.Z0 = self.units.Z0 if Z0 is None else Z0
        if dipole_moment is None:
            dipole_moment = constants.default_dipole_moment(self.units)
        source_x, source_y, source_z, source_pol = _check_source(source)
        _check_axis(x_pts, wavelength)
        _check_axis(y_pts, wavelength)
        S = lens_periphery_summary
        wl_nm = int(round(wavelength / self.units.nm))
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
        self.ux, self.uy = _lib.f64(np.ravel(ux)), _lib.f64(np.ravel(uy))
        self.pair_list = bool(pair_list)
        self.dxp = x_pts[1] - x_pts[0]
        self.dyp = y_pts[1] - y_pts[0]
        # rows of this rank.  INTERLEAVED (blocks of rows dealt round robin) when the direction grid
        # along x sits on the aperture's FFT lattice: the column pass then costs every rank 1 / world
        # of the whole aperture's and the ranks' loads balance by construction (metalens_hip.h
        # ml_farfield_interleave_block); else mirrored row pairs when the aperture has an even number
        # of rows (both far-field stages fold), one contiguous block otherwise (pair lists have no
        # folded form: csrc/farfield.hip rejects mirrored shards for them)
        self.interleave = 0
        if world > 1 and not self.pair_list and sharding in ('auto', 'interleaved'):
            self._plan()
            block = _lib.c_int(0)
            _lib.check(self.ctx.lib.ml_farfield_interleave_block(self.ctx.handle, world, _lib.byref(block)))
            self.interleave = block.value
            if sharding == 'interleaved' and not self.interleave:
                raise ValueError('this plan cannot be sharded by interleaved rows over %d ranks' % world)
        self.mirrored = (world > 1 and not self.pair_list and not self.interleave and
                         self.x_all.size % 2 == 0 and sharding != 'rows')
        self._shard_weights = None
        if self.interleave:
            self.row0 = self.row1 = 0
            self.rows = dist.interleaved_rows(self.x_all.size, world, rank, self.interleave)
        elif self.mirrored:
            # rows through the centre disc cost more near-field time (nearest-cell search, more
            # scattered table gathers); the GEMM cost per row is uniform.  Measured per-row totals
            # (tools/shard_probe.py, 8-way shards): rim 8.0e-4 ms, centre 8.7e-4 ms -> +13 % for
            # a row that lies entirely inside the centre disc
            r_c = float(S['r_min_list'][0])
            xs = self.x_all[:self.x_all.size // 2]
            chord = 2 * np.sqrt(np.maximum(r_c ** 2 - xs ** 2, 0.0))
            span = float(self.y[-1] - self.y[0]) or 1.0
            weights = 1.0 + 0.13 * np.minimum(chord / span, 1.0)
            self._shard_weights = weights
            self.row0, self.row1 = dist.mirrored_block(self.x_all.size, world, rank,
                                                       weights=weights)
            self.rows = dist.mirrored_rows(self.x_all.size, self.row0, self.row1)
        else:
            self.row0, self.row1 = dist.row_block(self.x_all.size, world, rank)
            self.rows = np.arange(self.row0, self.row1)
        self.sharding = ('interleaved blocks of %d rows' % self.interleave if self.interleave else
                         'mirrored pairs' if self.mirrored else 'rows' if world > 1 else 'whole aperture')
        self.x_local = np.ascontiguousarray(self.x_all[self.rows])
        # an empty shard would leave this rank without radiation vectors while the others wait
        # for it inside the all-reduce: refuse identically on every rank instead
        if world > 1:
            smallest = min(len(r) for r in (self._rows_of(k) for k in range(world)))
            if smallest == 0:
                raise ValueError('%d aperture rows cannot be sharded over %d ranks: a rank would '
                                 'hold no rows' % (self.x_all.size, world))
        self.shape = (self.ux.size,) if pair_list else (self.ux.size, self.uy.size)

    def _plan(self):
        ctx = self.ctx
        _lib.check(ctx.lib.ml_farfield_plan(ctx.handle, self.x_all.size, self.y.size, self.dxp,
                                            self.dyp, self.wavelength, self.n_glass,
                                            _lib.dptr(self.ux), self.ux.size, _lib.dptr(self.uy),
                                            self.uy.size, int(self.pair_list)))

    def _transform(self):
        """both transform stages of this rank's resident rows"""
        ctx, lib = self.ctx, self.ctx.lib
        if self.interleave:
            _lib.check(lib.ml_farfield_transform_interleaved_async(ctx.handle, self.interleave, self.world,
                                                                   self.rank, 0))
        elif self.mirrored:
            _lib.check(lib.ml_farfield_transform_mirrored_async(ctx.handle, self.row0, 0))
        else:
            _lib.check(lib.ml_farfield_transform_async(ctx.handle, self.row0, 0))

    def _rows_of(self, rank):
        """aperture rows rank ``rank`` owns under this object's partition"""
        n = self.x_all.size
        if self.interleave:
            return dist.interleaved_rows(n, self.world, rank, self.interleave)
        if self.mirrored:
            q0, q1 = dist.mirrored_block(n, self.world, rank, weights=self._shard_weights)
            return dist.mirrored_rows(n, q0, q1)
        r0, r1 = dist.row_block(n, self.world, rank)
        return np.arange(r0, r1)

    def set_source(self, source):
        """switch to another dipole / plane-wave source; tables, layout and grids stay resident"""
        source_x, source_y, source_z, source_pol = _check_source(source)
        self.params = nearfield_params(source_x, source_y, source_z, source_pol, self.wavelength,
                                       self.n_glass, self.dipole_moment, self.c0, self.Z0)

    def step_local(self):
        """near field + transform of this object's rows only (no reduction, no projection)"""
        ctx, lib = self.ctx, self.ctx.lib
        _lib.check(lib.ml_nearfield_premodulate(ctx.handle, int(self.fuse_modulation)))
        self._plan()
        if self.x_local.size:
            _lib.check(lib.ml_nearfield_async(ctx.handle, _lib.byref(self.params),
                                              _lib.dptr(self.x_local), self.x_local.size,
                                              _lib.dptr(self.y), self.y.size))
            self._transform()

    def queue_synthesis(self):
        """first half of a step: near field of this rank's rows into the resident field set"""
        ctx, lib = self.ctx, self.ctx.lib
        _lib.check(lib.ml_nearfield_premodulate(ctx.handle, int(self.fuse_modulation)))
        self._plan()
        if self.x_local.size:
            _lib.check(lib.ml_nearfield_async(ctx.handle, _lib.byref(self.params),
                                              _lib.dptr(self.x_local), self.x_local.size,
                                              _lib.dptr(self.y), self.y.size))

    def queue_transform(self):
        """second half: both transform stages, the reduction over ranks, the projection"""
        ctx, lib = self.ctx, self.ctx.lib
        if self.x_local.size:
            self._transform()
        if (self.world > 1 or dist.force_rccl()) and self.reduce == 'amplitudes':
            _lib.check(lib.ml_farfield_project_reduce(ctx.handle, self.Z0))
        else:
            if (self.world > 1 or dist.force_rccl()) and self.reduce == 'vectors':
                _lib.check(lib.ml_farfield_allreduce(ctx.handle))
            _lib.check(lib.ml_farfield_project_async(ctx.handle, self.Z0))   # ('none': this rank's partial sums)

    def step(self):
        """queue one pass of the hot path on the context's stream (asynchronous)"""
        self.queue_synthesis()
        self.queue_transform()

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
        # as build_nearfield does: up to three rounds (a settled tie can uncover another), the
        # answers so far carried along; anything still open after that is an error
        known = None
        for _ in range(3):
            now = (ties.settle(self.ctx, self._cells, self.x_local, self.y, known=known)
                   if self.x_local.size else None)
            redo = 0.0 if now is None else 1.0
            if self.world > 1 or dist.force_rccl():   # every rank repeats the pass or none does
                redo = float(dist.allreduce_host(self.ctx, [redo], 'max')[0])
            if not redo:
                break
            known = now if now is not None else known
            self.step()
            self.sync()
        else:
            # 'still open' is decided by ALL ranks together, as 'redo' was: a rank that raised alone
            # would leave the others waiting in _fetch's collective
            still = 1.0 if self.x_local.size and ties.pending(self.ctx).size else 0.0
            if self.world > 1 or dist.force_rccl():
                still = float(dist.allreduce_host(self.ctx, [still], 'max')[0])
            if still:
                raise RuntimeError('nearest-cell ties still open after three rounds'
                                   + ('' if self.x_local.size and ties.pending(self.ctx).size
                                      else ' (on another rank)'))
        return self._fetch()

    def _fetch(self):
        ctx, lib = self.ctx, self.ctx.lib
        power = _lib.c_double(0)
        viol = (_lib.BoundViolation * 8)()
        n_viol = _lib.c_int(0)
        _lib.check(lib.ml_nearfield_result(ctx.handle, _lib.byref(power), viol, 8,
                                           _lib.byref(n_viol)))
        # a sample outside the characterisation tables is an error on EVERY rank: the ranks that
        # did not see it must not walk into the next collective alone
        left = float(bool(n_viol.value))
        if self.world > 1 or dist.force_rccl():
            left = float(dist.allreduce_host(ctx, [left], 'max')[0])
        if n_viol.value:
            _raise_violation(viol[0], ctx, self.units.nm)
        if left:
            raise ValueError('a sample on another rank fell outside the characterisation tables '
                             '(that rank reports the value and the bound)')
        # the step's reduce-scatter left every rank with the sum of ITS block of direction rows: the
        # whole map on every rank is a collective of its own, paid here and not per step
        # (an older build of the library, METALENS_HIP_LIB for A/B runs, has no reduce-scatter and nothing to gather)
        if (self.world > 1 or dist.force_rccl()) and self.reduce == 'amplitudes' and hasattr(lib, 'ml_farfield_gather'):
            _lib.check(lib.ml_farfield_gather(ctx.handle))
        P = np.empty(self.shape)
        a_theta = np.empty(self.shape, dtype=np.complex128)
        a_phi = np.empty(self.shape, dtype=np.complex128)
        _lib.check(lib.ml_farfield_project(ctx.handle, self.Z0, _lib.dptr(P), _lib.dptr(a_theta),
                                           _lib.dptr(a_phi)))
        vec = [np.empty(self.shape, dtype=np.complex128) for _ in range(4)]
        _lib.check(lib.ml_farfield_download(ctx.handle, *[_lib.dptr(v) for v in vec]))
        local_power = power.value * self.dxp * self.dyp
        # 'tie_breaker': which scipy chose between exactly equidistant centre cells, if any sample needed it
        # (ties.settle: the reference's own tie-breaker is cKDTree's traversal order) - None: no ties on this grid
        return {'P': P, 'a_theta': a_theta, 'a_phi': a_phi, 'Nx': vec[0], 'Ny': vec[1],
                'Lx': vec[2], 'Ly': vec[3], 'power_local_rows': local_power,
                'tie_breaker': getattr(ctx, 'tie_settlement', None)}

