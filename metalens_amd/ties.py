"""Exact ties of the nearest-cell search, settled the way the reference settles them.

A centre-region sample that is exactly equidistant from two hexagonal cells - it sits on a
mirror line of the cell lattice, e.g. the x = 0 row of a symmetric grid with an odd number of
samples - has no defined nearest cell.  The reference takes whichever cell
``scipy.spatial.cKDTree.query`` returns (nearfield.py:363-364), which depends on how scipy
built its tree (sliding-midpoint splits, leaf order): only scipy can tell.  The HIP kernels
therefore report such samples (``ml_nearfield_ties``), this module asks cKDTree about exactly
those points - the same call on the same cell array as the reference makes - and hands the
answers back (``ml_nearfield_tie_answers``); the synthesis is then run again.  Ties depend on
the grid and the cells only, not on the source, so a sweep pays this once.  No ties (the usual
case: even sample counts) -> no scipy import, no second run.
"""
import warnings

import numpy as np

from . import _lib

_tree_cache = {}
# the scipy the golden fixtures of tests/golden were generated with (tests/golden/gen/make_golden.py META):
# the tie choice IS that library's tree traversal, so a different version may legitimately choose differently
FIXTURE_SCIPY = '1.15.3'
# what settled the ties of the last call: {'scipy': version, 'fixture_scipy': ..., 'samples': n} (None: no ties yet)
last_settlement = None
_warned = set()


def _tree(lens_center_summary, token):
    import scipy
    from scipy.spatial import cKDTree   # the reference's own tie-breaker
    if scipy.__version__ != FIXTURE_SCIPY and scipy.__version__ not in _warned:
        _warned.add(scipy.__version__)
        warnings.warn('exact nearest-cell ties are settled by scipy.spatial.cKDTree of scipy %s; the parity fixtures '
                      'of this package were generated with scipy %s, whose tree may order equidistant cells differently'
                      % (scipy.__version__, FIXTURE_SCIPY), RuntimeWarning, stacklevel=3)
    hit = _tree_cache.get(token)
    if hit is None:
        _tree_cache.clear()             # one lens at a time is the normal use
        hit = cKDTree(np.asarray(lens_center_summary, dtype=float)[:, 0:2])
        _tree_cache[token] = hit
    return hit


def pending(ctx):
    """sample ids (local row * ny + column) the last synthesis on ``ctx`` could not settle"""
    lib = ctx.lib
    n = _lib.c_int(0)
    _lib.check(lib.ml_nearfield_ties(ctx.handle, None, 0, _lib.byref(n)))
    if n.value == 0:
        return np.zeros(0, dtype=np.int64)
    if n.value > _lib.TIE_CAPACITY:
        raise _lib.MetalensHipError('%d samples are exactly equidistant from two centre cells; '
                                    'at most %d can be settled per grid' % (n.value, _lib.TIE_CAPACITY))
    ids = np.empty(n.value, dtype=np.int64)
    _lib.check(lib.ml_nearfield_ties(ctx.handle, ids.ctypes.data_as(_lib.POINTER(_lib.c_int64)),
                                     ids.size, _lib.byref(n)))
    return np.unique(ids[:n.value])


def settle(ctx, lens_center_summary, x_local, y_pts, known=None):
    """Ask cKDTree about the samples the last synthesis reported and give the kernels its answers
    (together with ``known`` = (ids, cells) settled earlier for the same grid and layout).
    Returns (ids, cells) now in force, or None if nothing was pending."""
    ids = pending(ctx)
    if ids.size == 0:
        return None
    ny = len(y_pts)
    pts = np.column_stack((np.asarray(x_local, dtype=float)[ids // ny],
                           np.asarray(y_pts, dtype=float)[ids % ny]))
    cells = _tree(lens_center_summary, ctx.layout_token).query(pts)[1].astype(np.int32)
    import scipy
    global last_settlement
    last_settlement = {'scipy': scipy.__version__, 'fixture_scipy': FIXTURE_SCIPY, 'samples': int(ids.size)}
    ctx.tie_settlement = dict(last_settlement)   # recorded next to the results of this context (HotPath.results())
    if known is not None and known[0].size:
        ids = np.concatenate((known[0], ids))
        cells = np.concatenate((known[1], cells))
        ids, first = np.unique(ids, return_index=True)
        cells = cells[first]
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    _lib.check(ctx.lib.ml_nearfield_tie_answers(
        ctx.handle, ids.ctypes.data_as(_lib.POINTER(_lib.c_int64)),
        cells.ctypes.data_as(_lib.POINTER(_lib.c_int32)), ids.size))
    return ids, cells
