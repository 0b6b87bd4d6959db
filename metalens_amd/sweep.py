"""Batched source sweeps with the sums kept on the GPU.

The reference's docstring describes the use (nearfield.py:69-73): an isotropic or Lambertian
emitter is the INCOHERENT sum of x-, y- and z-polarised dipoles, possibly at several positions -
i.e. many runs of near field -> far field whose POWERS are added, and whose efficiency is
``sum(total_P) / sum(power_passing_through_lens)`` (nearfield_farfield.py:74, nearfield.py:474-477).

What a sweep shares:

* tables, layout, far-field plan and the per-sample geometry records (ring, sector, rotated
  coordinates, nearest cell: csrc/nearfield_fast.hip) - they depend on grid and lens only;
* within a group of sources at ONE position that differ in polarisation (the x, y, z triple): the
  whole per-sample evaluation except the two weights the incident H enters with - such a group
  is ONE synthesis pass (``ml_nearfield_batch_async``) that leaves one resident field set per
  member;
* the sums: ``P_sum``, per-source ``total_P`` and encircled power are accumulated by a kernel
  right after each projection (``ml_farfield_accumulate``); only ``P_sum`` and two scalars per
  source cross PCIe, whatever the sweep's length.
"""
import numpy as np

from . import _lib, constants, packing, ties
from .grating import n_glass as tabulated_n_glass
from .nearfield import _check_axis, _raise_violation, nearfield_params
from .pipeline import _check_source

MAX_BATCH = 3          # members of a polarisation batch (csrc/nearfield_dev.h MAX_POL)
MAX_SLOTS = 4096       # ML_MAX_SWEEP_SLOTS


class SourceSweep:
    def __init__(self, wavelength, lens_periphery_summary, lens_center_summary, hexgridset,
                 x_pts, y_pts, ux, uy, dipole_moment=None, c0=None, Z0=None, ctx=None,
                 precision=None, method=None, units=None):
        self.ctx = ctx or _lib.default_context()
        self.ctx.set_precision(precision or 'f64')
        self.ctx.set_method(method or 'auto')
        self.units = constants.as_units(units)   # the caller's unit system (nearfield.build_nearfield); default SI
        self.c0 = self.units.c0 if c0 is None else c0
        self.Z0 = self.units.Z0 if Z0 is None else Z0
        if dipole_moment is None:
            dipole_moment = constants.default_dipole_moment(self.units)
        _check_axis(x_pts, wavelength)
        _check_axis(y_pts, wavelength)
        S = lens_periphery_summary
        wl_nm = int(round(wavelength / self.units.nm))
        n_glass = S['gratingcollection_list'][0].grating_list[0].n_glass
        if n_glass == 0:
            n_glass = tabulated_n_glass(wl_nm)
        self.n_glass, self.wavelength = n_glass, wavelength
        self._tables = (S['gratingcollection_list'], hexgridset, wl_nm)
        self._layout = (S, lens_center_summary)
        self._cells = lens_center_summary
        self.dipole_moment = dipole_moment
        self.x, self.y = _lib.f64(x_pts), _lib.f64(y_pts)
        self.ux, self.uy = _lib.f64(np.ravel(ux)), _lib.f64(np.ravel(uy))
        self.dxp, self.dyp = x_pts[1] - x_pts[0], y_pts[1] - y_pts[0]

    def prepare(self, check_content=True):
        """make tables, layout and the far-field plan resident (no-ops when they already are).
        ``check_content=False`` (``queue``): trust the content hashes of the last full ``prepare``
        as long as the context still holds the tables and layout it left there - hashing the
        caller's 30 MB of cells on every queued pass would make a sweep host-bound"""
        ctx, lib = self.ctx, self.ctx.lib
        mine = getattr(self, '_resident', None)
        if check_content or mine is None or mine != (ctx.tables_token, ctx.layout_token):
            packing.upload_tables(ctx, *self._tables)
            packing.upload_layout(ctx, *self._layout)
            self._resident = (ctx.tables_token, ctx.layout_token)
        _lib.check(lib.ml_nearfield_premodulate(ctx.handle, 0))
        _lib.check(lib.ml_farfield_plan(ctx.handle, self.x.size, self.y.size, self.dxp, self.dyp,
                                        self.wavelength, self.n_glass, _lib.dptr(self.ux),
                                        self.ux.size, _lib.dptr(self.uy), self.uy.size, 0))

    def queue(self, sources):
        """queue the whole sweep on the GPU and return without synchronising (benchmarks);
        tie settlement and the downloads are ``run``'s business.

        NOT content-checked: ``queue`` trusts the hashes taken by the last ``prepare()`` / ``run()``
        of this object.  An IN-PLACE edit of a table or of ``lens_center_summary`` made since then is
        not noticed and the pass runs on the resident (older) content - call ``prepare()`` (or
        ``run``) after editing.  Another object replacing the context's tables or layout IS
        noticed (the context's tokens change) and triggers a full re-check."""
        self.prepare(check_content=False)
        weights = np.ones(len(sources))
        for g in self._group(sources):
            self._pass(g, None, (0.0, 0.0, 0.0), weights)

    def _group(self, sources):
        """consecutive sources at one position -> batches of up to MAX_BATCH polarisations (one
        synthesis pass with everything but two weights per sample shared); then runs of
        single-source groups at DIFFERENT positions (a field-of-view sweep) -> position batches of up
        to MAX_BATCH members (synthesised back to back before any of them is transformed).  Members
        of a group carry their own position; plane waves and dipoles do not mix."""
        groups = []
        for k, src in enumerate(sources):
            sx, sy, sz, pol = _check_source(src)
            if groups and groups[-1]['pos'] == (sx, sy, sz) and len(groups[-1]['members']) < MAX_BATCH:
                groups[-1]['members'].append((k, pol))
                groups[-1]['positions'].append((sx, sy, sz))
            else:
                groups.append({'pos': (sx, sy, sz), 'members': [(k, pol)], 'positions': [(sx, sy, sz)]})
        merged = []
        for g in groups:
            last = merged[-1] if merged else None
            single = len(g['members']) == 1
            if (single and last is not None and last.get('mixed', len(last['members']) == 1)
                    and len(last['members']) < MAX_BATCH
                    and (last['positions'][0][2] == -float('inf')) == (g['pos'][2] == -float('inf'))):
                last['members'] += g['members']
                last['positions'] += g['positions']
                last['mixed'] = True
                last['pos'] = None
            else:
                merged.append(g)
        return merged

    def _pass(self, group, slots_done, cone, weights):
        """queue one group: batched synthesis, then per member transform -> projection -> sums"""
        ctx, lib = self.ctx, self.ctx.lib
        n = len(group['members'])
        params = (_lib.NearfieldParams * n)()
        for m, (k, pol) in enumerate(group['members']):
            sx, sy, sz = group['positions'][m]
            params[m] = nearfield_params(sx, sy, sz, pol, self.wavelength, self.n_glass,
                                         self.dipole_moment, self.c0, self.Z0)
        _lib.check(lib.ml_nearfield_batch_async(ctx.handle, params, n, _lib.dptr(self.x), self.x.size,
                                                _lib.dptr(self.y), self.y.size))
        # member by member, each transform's second stage reading what its first stage has just
        # written (134 MB at 4096^2 -> 512^2: it is still in the 256 MB memory-side cache).  Stacking
        # the members' 4 S planes through one stage-1 launch was measured and taken out again: stage 1
        # 0.535 -> 0.512 ms for x+y+z, but stage 2 then reads a 402 MB intermediate from HBM,
        # 0.171 -> 0.268 ms (experiment log of round 4)
        for m, (k, pol) in enumerate(group['members']):
            _lib.check(lib.ml_fields_select(ctx.handle, m))
            _lib.check(lib.ml_farfield_transform_async(ctx.handle, 0, 0))
            _lib.check(lib.ml_farfield_project_async(ctx.handle, self.Z0))
            _lib.check(lib.ml_farfield_accumulate(ctx.handle, float(weights[k]), cone[0], cone[1],
                                                  cone[2], k, int(k == 0)))
        return n

    def run(self, sources, weights=None, cone=None, cone_center=(0.0, 0.0), keep_each=False):
        """``sources`` = iterable of ``(source_x, source_y, source_z, source_pol)``; sources at the
        same position that follow each other (the x, y, z dipoles of one emitter) are synthesised
        together.  ``weights[k]`` scales source k in ``P_sum`` (default 1); ``cone`` = sine of the
        half-angle of the cone (about ``cone_center``) whose encircled power is wanted.

        Returns a dict: ``P_sum`` (incoherent sum of the far-field power maps, NaN outside the
        unit circle), ``power_in`` (incident power through the lens per source,
        nearfield.py:474-477), ``total_P`` (radiated power per source: sum of the finite
        ``P * dux * duy``, nearfield_farfield.py:74), ``efficiency`` = sum(total_P) /
        sum(power_in), ``cone_P`` / ``cone_efficiency`` if a cone was given, ``P_each`` if
        ``keep_each`` (downloads every map: for tests)."""
        sources = list(sources)
        if not 1 <= len(sources) <= MAX_SLOTS:
            raise ValueError('a sweep takes 1 to %d sources, got %d' % (MAX_SLOTS, len(sources)))
        weights = np.ones(len(sources)) if weights is None else np.asarray(weights, dtype=float)
        if weights.shape != (len(sources),):
            raise ValueError('weights must have one entry per source: got shape %s for %d sources'
                             % (weights.shape, len(sources)))
        ctx, lib = self.ctx, self.ctx.lib
        self.prepare()
        cone3 = (float(cone) if cone is not None else 0.0, float(cone_center[0]), float(cone_center[1]))
        groups = self._group(sources)
        power_in = np.zeros(len(sources))
        each = [None] * len(sources)
        first = True
        for g in groups:
            n = self._pass(g, None, cone3, weights)
            if first:
                # exact nearest-cell ties are a property of grid and cells: settled once, by
                # asking cKDTree like the reference (ties.py), then the pass is repeated
                # (as build_nearfield does: up to three rounds - a settled tie can uncover another -
                # with the answers so far carried along; anything left after that is an error)
                ctx.sync()
                known = None
                for _ in range(3):
                    known = ties.settle(ctx, self._cells, self.x, self.y, known=known)
                    if known is None:
                        break
                    n = self._pass(g, None, cone3, weights)
                    ctx.sync()
                else:
                    if ties.pending(ctx).size:
                        raise RuntimeError('nearest-cell ties still open after three rounds')
                first = False
            pw = np.zeros(n)
            _lib.check(lib.ml_nearfield_powers(ctx.handle, _lib.dptr(pw), n))
            viol = (_lib.BoundViolation * 8)()
            n_viol = _lib.c_int(0)
            _lib.check(lib.ml_nearfield_result(ctx.handle, None, viol, 8, _lib.byref(n_viol)))
            if n_viol.value:
                _raise_violation(viol[0], ctx, self.units.nm)
            for m, (k, pol) in enumerate(g['members']):
                power_in[k] = pw[m] * self.dxp * self.dyp
            if keep_each:   # the projections of this group's members, one at a time (tests)
                for m, (k, pol) in enumerate(g['members']):
                    _lib.check(lib.ml_fields_select(ctx.handle, m))
                    _lib.check(lib.ml_farfield_transform_async(ctx.handle, 0, 0))
                    P = np.empty((self.ux.size, self.uy.size))
                    _lib.check(lib.ml_farfield_project(ctx.handle, self.Z0, _lib.dptr(P), None, None))
                    each[k] = P
        P_sum = np.empty((self.ux.size, self.uy.size))
        total_P = np.zeros(len(sources))
        cone_P = np.zeros(len(sources))
        _lib.check(lib.ml_farfield_sums(ctx.handle, _lib.dptr(P_sum), _lib.dptr(total_P),
                                        _lib.dptr(cone_P), len(sources)))
        out = {'P_sum': P_sum, 'power_in': power_in, 'total_P': total_P}
        out['efficiency'] = total_P.sum() / power_in.sum() if power_in.sum() else np.nan
        if cone is not None:
            out['cone_P'] = cone_P
            out['cone_efficiency'] = cone_P.sum() / power_in.sum() if power_in.sum() else np.nan
        if keep_each:
            out['P_each'] = each
        return out


class WavelengthSweep:
    """Sources x WAVELENGTHS ("(source_x, source_y, source_pol, lambda) per
    launch"; the tri-colour emitter of BASELINE configs[3] on ONE GPU).

    A wavelength changes everything a context holds - the table key (nearfield.py:111), the sample
    grid (pitch lambda / 2.2, :95-97), hence the geometry records, the far-field plan - so each
    member wavelength gets a context of its own on the same GPU, and all of them stay resident
    side by side: tables, layout, records, field sets and plans are uploaded / built ONCE per
    wavelength, not once per (source, wavelength) visit as with one context re-used in turn (a
    4096^2 member is 1.2 GB of a 288 GB GPU).  The members' passes are queued on their own streams;
    nothing synchronises between them until the sums are fetched.

    ``members`` = one dict per wavelength with the arguments of ``SourceSweep`` (``wavelength``,
    ``lens_periphery_summary``, ``lens_center_summary``, ``hexgridset``, ``x_pts``, ``y_pts``,
    ``ux``, ``uy`` and optionally ``dipole_moment``, ``c0``, ``Z0``, ``units``); the lens objects may
    be shared between members (one collection list characterised at several wavelengths) or not."""

    def __init__(self, members, device=None, precision=None, method=None):
        if not members:
            raise ValueError('a wavelength sweep needs at least one member')
        self.contexts = [_lib.Context(device) for _ in members]
        self.sweeps = [SourceSweep(ctx=ctx, precision=precision, method=method, **m)
                       for ctx, m in zip(self.contexts, members)]
        self.wavelengths = [m['wavelength'] for m in members]

    def queue(self, sources):
        """queue every member's whole sweep and return without synchronising (benchmarks; see
        ``SourceSweep.queue`` for what is and is not re-checked)"""
        for sw in self.sweeps:
            sw.queue(sources)

    def sync(self):
        for ctx in self.contexts:
            ctx.sync()

    def run(self, sources, weights=None, spectrum=None, **kwargs):
        """``SourceSweep.run`` per wavelength (same sources, weights and cone).  Returns a dict:
        ``per_wavelength`` (the members' result dicts, in order) and ``efficiency`` = the spectrum-
        weighted ``sum_l s_l sum_k total_P / sum_l s_l sum_k power_in`` (``spectrum`` = relative
        spectral weights s_l, default equal); ``P_sum`` = the spectrum-weighted sum of the members'
        maps when they share one direction grid (else absent)."""
        sources = list(sources)
        s = np.ones(len(self.sweeps)) if spectrum is None else np.asarray(spectrum, dtype=float)
        if s.shape != (len(self.sweeps),):
            raise ValueError('spectrum must have one weight per wavelength')
        # (member after member: each run settles its own nearest-cell ties and reads its sums back;
        # what a member left on its context - tables, layout, records, plan - is there for the next call)
        out = [sw.run(sources, weights=weights, **kwargs) for sw in self.sweeps]
        res = {'per_wavelength': out, 'wavelengths': list(self.wavelengths)}
        num = sum(w * o['total_P'].sum() for w, o in zip(s, out))
        den = sum(w * o['power_in'].sum() for w, o in zip(s, out))
        res['efficiency'] = num / den if den else np.nan
        first = self.sweeps[0]
        if all(np.array_equal(sw.ux, first.ux) and np.array_equal(sw.uy, first.uy) for sw in self.sweeps):
            res['P_sum'] = sum(w * o['P_sum'] for w, o in zip(s, out))
        return res

    def close(self):
        for ctx in self.contexts:
            ctx.close()
