"""TEST INFRASTRUCTURE - CPU oracle, never imported by the product path.

Trilinear interpolation on a rectilinear grid with the exact arithmetic of
``scipy.interpolate.RegularGridInterpolator(method='linear')`` - the
third-party routine the reference calls at nearfield.py:310-311,424-425 through
grating.py:1227 / lens_center.py:222.  scipy is un-pinned by the reference (no
requirements file); this restates scipy 1.15.3 ``_rgi.py:_evaluate_linear`` and
``_rgi_cython.pyx:find_indices`` (SURVEY.md §8 a6'):

* per axis, cell ``i`` = largest index with ``grid[i] <= x``, clamped to
  ``[0, n-2]``; ``t = (x - grid[i]) / (grid[i+1] - grid[i])``;
* result = sum over the 8 cell corners, in ``itertools.product`` order (last
  axis fastest), of ``value * (((1*w0)*w1)*w2)`` with ``w = 1-t`` for the lower
  and ``t`` for the upper corner, accumulated left to right starting from 0.

Pinned by tests/golden/rgi_samples.npz (raw scipy outputs incl. exact-node and
upper-edge queries).
"""
import numpy as np


def locate(axis, x):
    i = np.searchsorted(axis, x, side='right') - 1
    i = np.minimum(np.maximum(i, 0), axis.size - 2)
    t = (x - axis[i]) / (axis[i + 1] - axis[i])
    return i, t


def trilinear(grid, values, p0, p1, p2):
    """``values[U,V,W]`` (any dtype) sampled at the points ``(p0[j], p1[j], p2[j])``."""
    (i0, t0), (i1, t1), (i2, t2) = (locate(np.asarray(g, dtype=float), np.asarray(p, dtype=float))
                                    for g, p in zip(grid, (p0, p1, p2)))
    acc = np.zeros(np.shape(p0), dtype=np.result_type(values, float))
    for c0, w0 in ((0, 1 - t0), (1, t0)):
        for c1, w1 in ((0, 1 - t1), (1, t1)):
            for c2, w2 in ((0, 1 - t2), (1, t2)):
                weight = ((1.0 * w0) * w1) * w2
                acc = acc + values[i0 + c0, i1 + c1, i2 + c2] * weight
    return acc
