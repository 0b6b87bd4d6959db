"""Flatten the reference's duck-typed inputs into the contiguous arrays the C ABI
takes and upload them.

Accepted objects are whatever the reference's ``build_nearfield`` accepts
(nearfield.py:87-93,111,264,310,390-392): a ``lens_periphery_summary`` dict, a
``lens_center_summary`` ``[C,3]`` array, and collection / HexGridSet objects
exposing ``.grating_list[i].data``, ``.grating_list[0].n_glass / .grating_period
/ .lateral_period``, ``.interpolators[(wavelength_in_nm, (ox, oy), 'x'|'y',
'ampfy'|'ampfx')]`` (anything with ``.grid`` and ``.values``: a scipy
``RegularGridInterpolator`` or this package's ``TrilinearTable``) and
``.interpolator_bounds``.
"""
import hashlib
from math import pi

import os

import numpy as np

from . import _lib

MAX_SLOTS = 32
MAX_ORDERS = 32
POL_AMP = (('x', 'ampfy'), ('x', 'ampfx'), ('y', 'ampfy'), ('y', 'ampfx'))


def orders_of(obj):
    """the union of diffraction orders over the object's records, sorted
    (the reference iterates a set, nearfield.py:264,390 - order only matters to rounding)"""
    return sorted({(e['ox'], e['oy']) for g in obj.grating_list for e in g.data})


def pack_table(obj, wavelength_in_nm):
    """-> dict of contiguous arrays for ml_upload_table.  ``KeyError`` if the
    object has no table for this wavelength / order, as in the reference."""
    orders = orders_of(obj)
    if not 1 <= len(orders) <= MAX_ORDERS:
        raise ValueError('a table needs between 1 and %d diffraction orders, got %d'
                         % (MAX_ORDERS, len(orders)))
    first = obj.interpolators[(wavelength_in_nm, orders[0], 'x', 'ampfy')]
    axes = [np.ascontiguousarray(g, dtype=np.float64) for g in first.grid]
    shape = tuple(a.size for a in axes)
    values = np.empty((len(orders),) + shape + (4,), dtype=np.complex128)
    for o, order in enumerate(orders):
        for q, (pol, amp) in enumerate(POL_AMP):
            f = obj.interpolators[(wavelength_in_nm, order, pol, amp)]
            if tuple(len(g) for g in f.grid) != shape or any(
                    not np.array_equal(np.asarray(g, dtype=float), a) for g, a in zip(f.grid, axes)):
                raise ValueError('interpolators of one collection must share one grid')
            values[o, :, :, :, q] = np.asarray(f.values)
    bounds = np.zeros(6)
    b = obj.interpolator_bounds
    bounds[:len(b)] = [float(v) for v in b][:6]
    return {'axes': axes, 'orders': np.ascontiguousarray(orders, dtype=np.int32).reshape(-1, 2),
            # ox * 2*pi exactly as written at nearfield.py:268-269,395-396
            'order_k': np.array([[ox * 2 * pi, oy * 2 * pi] for ox, oy in orders], dtype=np.float64),
            'values': values, 'bounds': bounds}


try:                      # 128-bit content hash at memory speed (15x blake2b); both images have it
    import xxhash

    def _hasher():
        return xxhash.xxh3_128()
except ImportError:       # pragma: no cover
    def _hasher():
        return hashlib.blake2b(digest_size=16)


_POOL = None
_BIG = 4 << 20        # arrays beyond this are hashed in parallel chunks
_CHUNKS = 8


def _cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:    # pragma: no cover
        return os.cpu_count() or 1


def _chunk_digest(view):
    h = _hasher()
    h.update(view)
    return h.digest()


def _drop_pool():
    """a forked child inherits _POOL without its worker threads: start over there"""
    global _POOL
    _POOL = None


if hasattr(os, 'register_at_fork'):
    os.register_at_fork(after_in_child=_drop_pool)


def _feed(h, a):
    """hash an array's dtype, shape and bytes (no copy for contiguous arrays).  Large arrays (the
    cell list of a millimetre lens is 30 MB, hashed on EVERY drop-in call) are hashed as eight
    chunk digests fed in order.  The chunk layout depends on the byte count ALONE, so the token of
    a given content is the same on every host and under every CPU affinity; only WHERE the chunks
    are hashed differs (a thread pool when there are cores for it - xxhash releases the GIL)"""
    global _POOL
    a = np.ascontiguousarray(a)
    h.update(('%s%s' % (a.dtype.str, a.shape)).encode())
    view = memoryview(a).cast('B')
    if view.nbytes < _BIG:
        h.update(view)
        return
    step = -(-view.nbytes // _CHUNKS)
    chunks = [view[k:k + step] for k in range(0, view.nbytes, step)]
    if _cores() < 2 * _CHUNKS:
        digests = map(_chunk_digest, chunks)
    else:
        if _POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _POOL = ThreadPoolExecutor(max_workers=_CHUNKS)
        digests = _POOL.map(_chunk_digest, chunks)
    for d in digests:
        h.update(d)


def _tables_fingerprint(objs, wavelength_in_nm):
    """Content hash of what pack_table WOULD pack, taken straight from the caller's objects:
    every interpolator of this wavelength (grid + values), the bounds, the periods of the centre
    set.  Lets a repeated call skip the packing; any edit of a table changes it."""
    h = _hasher()
    for obj in objs:
        if obj is None:
            h.update(b'none')
            continue
        keys = sorted((k for k in obj.interpolators if k[0] == wavelength_in_nm), key=repr)
        for k in keys:
            f = obj.interpolators[k]
            h.update(repr(k).encode())
            for g in f.grid:
                _feed(h, np.asarray(g, dtype=np.float64))
            _feed(h, np.asarray(f.values))
        _feed(h, np.asarray([float(v) for v in obj.interpolator_bounds], dtype=np.float64))
        g0 = obj.grating_list[0]
        _feed(h, np.asarray([g0.grating_period, g0.lateral_period], dtype=np.float64))
        # the ORDER SET comes from the records, not from the interpolator keys (nearfield.py:264)
        h.update(repr(orders_of(obj)).encode())
    return h.digest()


def _digest(t):
    """content hash of one packed table (axes, orders, values, bounds)"""
    h = _hasher()
    for a in t['axes'] + [t['orders'], t['order_k'], t['values'], t['bounds']]:
        _feed(h, a)
    return h.digest()


def upload_tables(ctx, gratingcollection_list, hexgridset, wavelength_in_nm):
    if len(gratingcollection_list) > MAX_SLOTS:
        raise ValueError('at most %d grating collections per lens' % MAX_SLOTS)
    # compare CONTENT with what is already on the GPU: object identity is not a safe cache key
    # (tables can be edited in place, ids are reused after garbage collection).  First a hash of
    # the caller's own arrays - a repeated call stops here without packing anything -
    fp = _tables_fingerprint(list(gratingcollection_list) + [hexgridset], wavelength_in_nm)
    if getattr(ctx, 'tables_fingerprint', None) == fp and ctx.tables_token is not None:
        return
    # then, after packing, of the packed tables
    packed = [pack_table(gc, wavelength_in_nm) for gc in gratingcollection_list]
    center = pack_table(hexgridset, wavelength_in_nm) if hexgridset is not None else None
    periods = None
    if center is not None:
        g0 = hexgridset.grating_list[0]
        periods = np.array([g0.grating_period, g0.lateral_period], dtype=np.float64)
    token = ('tables', wavelength_in_nm, tuple(_digest(t) for t in packed),
             _digest(center) if center is not None else None,
             periods.tobytes() if periods is not None else None)
    if ctx.tables_token == token:
        ctx.tables_fingerprint = fp
        return
    lib = ctx.lib
    # nothing is known about the GPU's tables until every upload below has succeeded: a failure
    # part-way must not leave a fingerprint / token that a later call would take for "resident"
    ctx.tables_fingerprint = None
    ctx.tables_token = None
    for slot, t in enumerate(packed):
        a0, a1, a2 = t['axes']
        _lib.check(lib.ml_upload_table(
            ctx.handle, slot, _lib.dptr(a0), a0.size, _lib.dptr(a1), a1.size, _lib.dptr(a2), a2.size,
            _lib.iptr(t['orders']), _lib.dptr(t['order_k']), len(t['orders']),
            _lib.dptr(t['values']), _lib.dptr(t['bounds']), None))
    if center is not None:
        t = center
        a0, a1, a2 = t['axes']
        _lib.check(lib.ml_upload_table(
            ctx.handle, -1, _lib.dptr(a0), a0.size, _lib.dptr(a1), a1.size, _lib.dptr(a2), a2.size,
            _lib.iptr(t['orders']), _lib.dptr(t['order_k']), len(t['orders']),
            _lib.dptr(t['values']), _lib.dptr(t['bounds']), _lib.dptr(periods)))
    ctx.tables_token = token
    ctx.tables_fingerprint = fp
    ctx.table_orders = [p['orders'] for p in packed] + ([center['orders']] if center is not None else [])


def pack_layout(lens_periphery_summary, lens_center_summary):
    S = lens_periphery_summary
    r_min = _lib.f64(S['r_min_list'])
    r_max = _lib.f64(S['r_max_list'])
    r_center = _lib.f64(S['r_center_list'])
    period = _lib.f64(S['grating_period_list'])
    num = np.asarray(S['num_around_circle_list'])
    # per-ring constants, evaluated exactly like the reference's per-sample expressions
    # (nearfield.py:125,163,167)
    boundaries = np.ascontiguousarray(np.hstack((r_min, r_max[-1])))
    dphi = _lib.f64(2 * pi / num)
    lateral = _lib.f64(r_center * dphi)
    ring_gc = np.ascontiguousarray(S['gratingcollection_index_here_list'], dtype=np.int32)
    # (cos, sin) of every possible grating rotation sector*dphi (nearfield.py:169-171), one
    # run of the table per distinct dphi.  Evaluated HERE with NumPy - not on the GPU - because
    # the rotation multiplies lens-sized coordinates inside phases of ~1e4 rad: one ulp of
    # cos/sin is ~1e-12 rad of phase, and this way it is the same ulp the reference gets.
    #
    # The sector itself is round(arctan2(y, x) / dphi).  A symmetric sample grid puts samples on
    # (or one ulp off) the diagonals, where phi / dphi can be an exact tie, and then the decision
    # hangs on the last bit of arctan2.  For such near-tie samples the kernel recomputes phi to
    # ~1e-19 from the boundary angle (k + 1/2) * dphi, whose value, cosine and sine are tabulated
    # here in extended precision, and rounds it - i.e. it uses the correctly rounded arctan2,
    # which is what NumPy returns on these arguments.
    rot_center = np.zeros(r_center.size, dtype=np.int32)
    rot_half = np.zeros(r_center.size, dtype=np.int32)
    runs, ties, at = [], [], 0
    for value in np.unique(dphi):
        half = int(np.ceil(pi / value)) + 1
        sector = np.arange(-half - 1, half + 1, dtype=np.float64)   # entry k: sector k
        rot = sector * value
        runs.append(np.column_stack((np.cos(rot), np.sin(rot))))
        b = (sector.astype(np.longdouble) + np.longdouble(0.5)) * np.longdouble(value)
        tie = np.empty((sector.size, 6))                            # entry k: boundary k + 1/2
        for col, ext in enumerate((b, np.cos(b), np.sin(b))):
            hi = ext.astype(np.float64)
            tie[:, 2 * col] = hi
            tie[:, 2 * col + 1] = (ext - hi.astype(np.longdouble)).astype(np.float64)
        ties.append(tie)
        rot_center[dphi == value] = at + half + 1
        rot_half[dphi == value] = half
        at += sector.size
    rot_table = np.ascontiguousarray(np.vstack(runs))
    tie_table = np.ascontiguousarray(np.vstack(ties))
    if lens_center_summary is None or len(lens_center_summary) == 0:
        cells = np.zeros((0, 3))
    else:
        cells = _lib.f64(np.asarray(lens_center_summary)[:, 0:3])
    return {'boundaries': boundaries, 'r_center': r_center, 'period': period, 'dphi': dphi,
            'lateral': lateral, 'ring_gc': ring_gc, 'cells': cells, 'rot_table': rot_table,
            'tie_table': tie_table,
            'rot_center': rot_center, 'rot_half': rot_half}


def upload_layout(ctx, lens_periphery_summary, lens_center_summary):
    # content hash over EVERY input array (a cheaper checksum - shape + sums - cannot see a
    # re-ordered cell list or a swap of two cell types, and tie answers (ties.py) are row indices
    # into exactly this cell order), taken BEFORE packing: a repeated call stops here
    S = lens_periphery_summary
    h = _hasher()
    for key in ('r_min_list', 'r_max_list', 'r_center_list', 'grating_period_list',
                'num_around_circle_list', 'gratingcollection_index_here_list'):
        h.update(key.encode())
        _feed(h, np.asarray(S[key]))
    if lens_center_summary is not None and len(lens_center_summary):
        _feed(h, np.asarray(lens_center_summary))
    token = ('layout', h.digest())
    if ctx.layout_token == token:
        return
    L = pack_layout(lens_periphery_summary, lens_center_summary)
    n_rings = L['r_center'].size
    _lib.check(ctx.lib.ml_upload_layout(
        ctx.handle, n_rings, _lib.dptr(L['boundaries']), _lib.dptr(L['r_center']),
        _lib.dptr(L['period']), _lib.dptr(L['dphi']), _lib.dptr(L['lateral']),
        _lib.iptr(L['ring_gc']), _lib.dptr(L['rot_table']), _lib.dptr(L['tie_table']),
        len(L['rot_table']),
        _lib.iptr(L['rot_center']), _lib.iptr(L['rot_half']), len(L['cells']),
        _lib.dptr(L['cells']) if len(L['cells']) else None))
    ctx.layout_token = token
