"""TEST INFRASTRUCTURE - CPU oracle, never imported by the product path.

NumPy restatement of the reference's near-to-far-field transform
(reference nearfield_farfield.py:14-191) and of the aperture -> direction sum
its docstring derives (nearfield_farfield.py:97-138), which is what the
MI355X build evaluates for an arbitrary M x M' direction grid instead of the
caller-side ``fft2(fftshift(F))`` (nearfield_farfield.py:18-20).

Conventions restated from the reference:

* equivalent currents  Jx=-Hy, Jy=Hx, Mx=Ey, My=-Ex  (nearfield_farfield.py:91-93)
  so  Nx=-F[Hy] dA, Ny=F[Hx] dA, Lx=F[Ey] dA, Ly=-F[Ex] dA  (:135-138);
* F[.](ux,uy) = sum_{m1,m2} Fshift[m1,m2] exp(-i k (m1 dx ux + m2 dy uy)),
  k = 2 pi n_glass / wavelength (:111-120).  ``fftshift`` moves original sample
  ``j`` to ``(j + n//2) mod n``, so on the FFT lattice the sum equals
  sum_j F[j] exp(-i k (j - ceil(n/2)) dx ux); we take that form (sample
  ``ceil(n/2)`` is the phase origin) for every direction, on or off the lattice;
* direction cosines of FFT bin i: ``i * (wavelength/n_glass) / (dx * n)``,
  wrapped by ``-(wavelength/n_glass)/dx`` above half the maximum (:35-39);
* projection, NaN rule, +1e-9 / +1e-5 regularisers, the (0,0) special case and
  the final factor 2 (:153-189).
"""
from math import pi

import numpy as np

Z0_DEFAULT = 1.25663706212e-6 * 299792458.0


def _check_axis(pts, wavelength):
    d = np.diff(np.asarray(pts, dtype=float))
    assert 0 < d[0] < wavelength / 2
    assert d.max() - d.min() <= 1e-9 * np.abs(d).max()


def fft_direction_cosines(n, step, wavelength, n_glass):
    """un-shifted direction cosines of the n FFT bins (nearfield_farfield.py:35-39)"""
    u = np.arange(n) * (wavelength / n_glass) / (step * n)
    u[u > u.max() / 2] -= (wavelength / n_glass) / step
    return u


def project(Nx, Ny, Lx, Ly, ux, uy, wavelength, n_glass, Z0=Z0_DEFAULT,
            return_amplitudes=False):
    """Far-field power per unit (dux duy), ``P r^2 / uz``, from the radiation
    vectors on a tensor grid ``ux[:,None]``, ``uy[None,:]``
    (nearfield_farfield.py:153-189).  With ``return_amplitudes`` also returns the
    two complex far-field amplitudes ``L_phi + Z N_theta`` (prop. to E_theta) and
    ``L_theta - Z N_phi`` (prop. to -E_phi), which the reference computes but
    never returns."""
    ux = np.asarray(ux, dtype=float).reshape(-1, 1)
    uy = np.asarray(uy, dtype=float).reshape(1, -1)
    uz = 1 - ux ** 2 - uy ** 2
    uz[uz < 0] = np.nan
    uz = uz ** 0.5
    sintheta = (ux ** 2 + uy ** 2) ** 0.5
    Nth = Nx * ux * uz / (sintheta + 1e-9) + Ny * uy * uz / (sintheta + 1e-9)
    Nph = -Nx * uy / (sintheta + 1e-9) + Ny * ux / (sintheta + 1e-9)
    i = np.where(ux == 0)[0]
    j = np.where(uy == 0)[1]
    if i.size and j.size:
        Nth[np.ix_(i, j)] = Nx[np.ix_(i, j)]
        Nph[np.ix_(i, j)] = Ny[np.ix_(i, j)]
    Lth = Lx * ux * uz / (sintheta + 1e-9) + Ly * uy * uz / (sintheta + 1e-9)
    Lph = -Lx * uy / (sintheta + 1e-9) + Ly * ux / (sintheta + 1e-9)
    if i.size and j.size:
        Lth[np.ix_(i, j)] = Lx[np.ix_(i, j)]
        Lph[np.ix_(i, j)] = Ly[np.ix_(i, j)]
    Z = Z0 / n_glass
    a_theta = Lph + Z * Nth
    a_phi = Lth - Z * Nph
    P = ((2 * pi * n_glass / wavelength) ** 2 / (32 * pi ** 2 * Z)
         * (abs(a_theta) ** 2 + abs(a_phi) ** 2)) / (uz + 1e-5)
    P *= 2
    if return_amplitudes:
        return P, a_theta, a_phi
    return P


def farfield_from_nearfield(fftEx, fftEy, fftHx, fftHy, xp_list, yp_list, wavelength,
                            n_glass, Z0=Z0_DEFAULT):
    """Same signature and return tuple as the reference
    (nearfield_farfield.py:14,75) plus explicit ``Z0``."""
    dxp = xp_list[1] - xp_list[0]
    dyp = yp_list[1] - yp_list[0]
    nx, ny = len(xp_list), len(yp_list)
    assert fftEx.shape == fftEy.shape == fftHx.shape == fftHy.shape == (nx, ny)
    _check_axis(xp_list, wavelength)
    _check_axis(yp_list, wavelength)
    ux = fft_direction_cosines(nx, dxp, wavelength, n_glass)
    uy = fft_direction_cosines(ny, dyp, wavelength, n_glass)
    Nx = -fftHy * dxp * dyp
    Ny = fftHx * dxp * dyp
    Lx = fftEy * dxp * dyp
    Ly = -fftEx * dxp * dyp
    P = project(Nx, Ny, Lx, Ly, ux, uy, wavelength, n_glass, Z0)
    P = np.fft.fftshift(P)
    ux = np.fft.fftshift(ux)
    uy = np.fft.fftshift(uy)
    dux = ux[1] - ux[0]
    duy = uy[1] - uy[0]
    total_P = (P * dux * duy)[np.isfinite(P)].sum()
    return P, total_P, ux.reshape(-1, 1), uy.reshape(1, -1), dux, duy


def axis_twiddles(n, step, u, wavelength, n_glass):
    """``T[a, j] = exp(-i k (j - ceil(n/2)) step u[a])`` with the phase reduced
    mod 1 turn in extended precision (np.longdouble) before the sin/cos."""
    u = np.asarray(u, dtype=np.longdouble).reshape(-1, 1)
    pos = (np.arange(n) - (n - n // 2)).astype(np.longdouble).reshape(1, -1) * np.longdouble(step)
    turns = pos * u * (np.longdouble(n_glass) / np.longdouble(wavelength))
    turns = turns - np.rint(turns)
    ang = (-2 * np.longdouble(pi)) * turns
    return (np.cos(ang) + 1j * np.sin(ang)).astype(complex)


def radiation_vectors(Ex, Ey, Hx, Hy, xp_list, yp_list, wavelength, n_glass, ux, uy,
                      row_range=None):
    """``Nx, Ny, Lx, Ly`` on the tensor grid ``ux x uy`` by direct summation
    over the aperture, written as two dense products A @ F @ B^T
    (nearfield_farfield.py:111-120,135-138).  With ``row_range=(r0, r1)`` the fields
    hold only aperture rows r0..r1-1 of ``xp_list`` and the result is that block's
    partial sum (the multi-GPU decomposition)."""
    dxp = xp_list[1] - xp_list[0]
    dyp = yp_list[1] - yp_list[0]
    A = axis_twiddles(len(xp_list), dxp, ux, wavelength, n_glass)
    if row_range is not None:
        # (r0, r1) = a contiguous block; an index array = any subset of rows (mirrored shards)
        if len(row_range) == 2 and np.ndim(row_range[0]) == 0:
            A = A[:, row_range[0]:row_range[1]]
        else:
            A = A[:, np.asarray(row_range)]
    B = axis_twiddles(len(yp_list), dyp, uy, wavelength, n_glass)
    dA = dxp * dyp

    def transform(F):
        return (A @ np.asarray(F, dtype=complex)) @ B.T

    return (-transform(Hy) * dA, transform(Hx) * dA, transform(Ey) * dA, -transform(Ex) * dA)


def radiation_vectors_pairs(Ex, Ey, Hx, Hy, xp_list, yp_list, wavelength, n_glass, ux, uy):
    """Same sum for an arbitrary LIST of directions ``(ux[d], uy[d])`` (no
    tensor structure).  O(D * N^2): small cases only."""
    dxp = xp_list[1] - xp_list[0]
    dyp = yp_list[1] - yp_list[0]
    A = axis_twiddles(len(xp_list), dxp, ux, wavelength, n_glass)   # [D, nx]
    B = axis_twiddles(len(yp_list), dyp, uy, wavelength, n_glass)   # [D, ny]
    dA = dxp * dyp

    def transform(F):
        return np.einsum('dx,xy,dy->d', A, np.asarray(F, dtype=complex), B)

    return (-transform(Hy) * dA, transform(Hx) * dA, transform(Ey) * dA, -transform(Ex) * dA)


def farfield_direct(Ex, Ey, Hx, Hy, xp_list, yp_list, wavelength, n_glass, ux, uy,
                    Z0=Z0_DEFAULT):
    """Radiation vectors + projection on an M x M' direction grid."""
    Nx, Ny, Lx, Ly = radiation_vectors(Ex, Ey, Hx, Hy, xp_list, yp_list, wavelength,
                                       n_glass, ux, uy)
    P, a_theta, a_phi = project(Nx, Ny, Lx, Ly, ux, uy, wavelength, n_glass, Z0,
                                return_amplitudes=True)
    return {'Nx': Nx, 'Ny': Ny, 'Lx': Lx, 'Ly': Ly, 'P': P, 'a_theta': a_theta, 'a_phi': a_phi}
