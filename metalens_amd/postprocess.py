"""Power bookkeeping after the far-field projection.

The reference stops at ``total_P`` (nearfield_farfield.py:74: the sum of the finite
``P * dux * duy``) and leaves the rest to the caller; its docstrings describe what callers do
with it (nearfield.py:69-73: an isotropic or Lambertian emitter is the incoherent sum of x-, y-
and z-polarised dipoles; efficiency = far-field power / ``power_passing_through_lens``).  These
are host-side reductions over an M x M map (a few hundred kilobytes): NumPy, no kernel.

``P`` is power per unit (ux, uy) in glass, NaN outside the unit circle, on the tensor grid
``ux[:, None], uy[None, :]`` (what ``farfield_from_nearfield`` and ``FarfieldTransform`` return).
"""
import numpy as np


def total_power(P, dux, duy):
    """sum of the finite ``P * dux * duy`` (nearfield_farfield.py:74)"""
    cell = np.asarray(P, dtype=float) * dux * duy
    return float(cell[np.isfinite(cell)].sum())


def _sin2(ux, uy):
    ux = np.asarray(ux, dtype=float).reshape(-1, 1)
    uy = np.asarray(uy, dtype=float).reshape(1, -1)
    return ux * ux + uy * uy


def encircled_power(P, ux, uy, dux, duy, half_angle=None, sin_max=None, center=(0.0, 0.0)):
    """Power radiated into the cone of half-angle ``half_angle`` [rad] (or direction-cosine
    radius ``sin_max``) about the direction ``center`` = (ux0, uy0): the sum of the finite
    ``P * dux * duy`` over grid points with (ux-ux0)^2 + (uy-uy0)^2 <= sin_max^2."""
    assert (half_angle is None) != (sin_max is None), 'give half_angle or sin_max'
    if sin_max is None:
        sin_max = np.sin(half_angle)
    cell = np.asarray(P, dtype=float) * dux * duy
    inside = _sin2(np.asarray(ux, dtype=float).ravel() - center[0],
                   np.asarray(uy, dtype=float).ravel() - center[1]) <= sin_max * sin_max
    return float(cell[inside & np.isfinite(cell)].sum())


def encircled_power_curve(P, ux, uy, dux, duy, sin_list, center=(0.0, 0.0)):
    """``encircled_power`` for an ascending list of cone radii, one pass over the map"""
    sin_list = np.asarray(sin_list, dtype=float)
    assert np.all(np.diff(sin_list) >= 0)
    cell = np.asarray(P, dtype=float) * dux * duy
    rho2 = _sin2(np.asarray(ux, dtype=float).ravel() - center[0],
                 np.asarray(uy, dtype=float).ravel() - center[1])
    ok = np.isfinite(cell)
    order = np.argsort(rho2[ok], kind='stable')
    cum = np.concatenate(([0.0], np.cumsum(cell[ok][order])))
    return cum[np.searchsorted(rho2[ok][order], sin_list * sin_list, side='right')]


def incoherent_sum(P_maps):
    """Far-field power map of an incoherent source: the plain sum of the maps of its mutually
    incoherent components (x, y and z dipoles of an isotropic emitter, nearfield.py:69-73;
    several wavelengths or positions).  NaN (outside the unit circle) stays NaN."""
    P_maps = [np.asarray(P, dtype=float) for P in P_maps]
    assert P_maps and all(P.shape == P_maps[0].shape for P in P_maps)
    return np.sum(P_maps, axis=0)


def efficiency(P, ux, uy, dux, duy, power_in, half_angle=None, sin_max=None, center=(0.0, 0.0)):
    """Fraction of the power that passed through the lens (``power_passing_through_lens`` of
    ``build_nearfield``, nearfield.py:474-477; summed over the sources for an incoherent sum)
    that ends up in the far field - in the whole map, or inside a cone if one is given."""
    power_in = float(np.sum(power_in))
    if half_angle is None and sin_max is None:
        out = total_power(P, dux, duy)
    else:
        out = encircled_power(P, ux, uy, dux, duy, half_angle, sin_max, center)
    return out / power_in if power_in else float('nan')
