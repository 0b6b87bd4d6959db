"""Regular-grid trilinear table: the host-side container for one packed
diffraction-amplitude grid.

The reference stores each grid in a ``scipy.interpolate.RegularGridInterpolator``
(reference grating.py:1227, lens_center.py:222) and the near-field code only
ever (a) calls it on an ``[n,3]`` point array and (b) we, the packer, read its
``.grid`` and ``.values``.  ``TrilinearTable`` offers exactly those three things
so that ``GratingCollection.interpolators[...]`` built by this package and by the
reference are interchangeable as far as the hot path is concerned.

``__call__`` is a convenience for host-side spot checks of a table; the GPU
near-field kernel never goes through it (it reads the packed arrays, see
packing.py).  Its arithmetic follows scipy 1.15 ``_evaluate_linear`` /
``find_indices`` (SURVEY.md §8 a6'): cell index = largest i with
``grid[i] <= x`` clamped to ``n-2``; corner weights multiplied as
``((1*w0)*w1)*w2`` and the eight terms added in ``itertools.product`` order.
"""
import numpy as np


class TrilinearTable:
    def __init__(self, grid, values, bounds_error=True):
        self.grid = tuple(np.ascontiguousarray(g, dtype=float) for g in grid)
        self.values = np.ascontiguousarray(values)
        if self.values.ndim != 3 or len(self.grid) != 3:
            raise ValueError('TrilinearTable is 3-D only')
        for ax, g in enumerate(self.grid):
            if g.ndim != 1 or g.size != self.values.shape[ax]:
                raise ValueError('grid/values shape mismatch on axis %d' % ax)
            if g.size < 2:
                raise ValueError('need at least two nodes on axis %d' % ax)
            if not np.all(np.diff(g) > 0):
                raise ValueError('axis %d is not strictly ascending' % ax)
        self.bounds_error = bounds_error

    @staticmethod
    def _locate(g, x):
        i = np.searchsorted(g, x, side='right') - 1
        np.clip(i, 0, g.size - 2, out=i)
        return i, (x - g[i]) / (g[i + 1] - g[i])

    def __call__(self, xi):
        xi = np.asarray(xi, dtype=float)
        if xi.shape[-1] != 3:
            raise ValueError('requested xi has dimension %d, need 3' % xi.shape[-1])
        pts = xi.reshape(-1, 3)
        if self.bounds_error:
            for ax, g in enumerate(self.grid):
                p = pts[:, ax]
                if p.size and not (np.all(p >= g[0]) and np.all(p <= g[-1])):
                    raise ValueError('One of the requested xi is out of bounds '
                                     'in dimension %d' % ax)
        loc = [self._locate(g, pts[:, ax]) for ax, g in enumerate(self.grid)]
        out = np.zeros(pts.shape[0], dtype=np.result_type(self.values, float))
        for c0 in (0, 1):
            for c1 in (0, 1):
                for c2 in (0, 1):
                    w = np.ones(pts.shape[0])
                    for c, (i, t) in zip((c0, c1, c2), loc):
                        w = w * (t if c else 1 - t)
                    v = self.values[loc[0][0] + c0, loc[1][0] + c1, loc[2][0] + c2]
                    out = out + v * w
        return out.reshape(xi.shape[:-1])
