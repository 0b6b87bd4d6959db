"""Stand-in for the absent third-party ``numericalunits`` package, used ONLY by
tests/golden/gen/make_golden.py to import the reference in the build container.

It exposes the handful of names the reference reads, in plain SI (what the real
package gives after ``reset_units('SI')``).  The values of c0 / Z0 enter the
reference's arithmetic; they are identical to
metalens_amd.constants and are recorded inside every fixture.
This file is written for this repo; it is not reference source.
"""
m = 1.0
cm = 1e-2
mm = 1e-3
um = 1e-6
nm = 1e-9
s = 1.0
kg = 1.0
C = 1.0
V = 1.0
A = 1.0
c0 = 299792458.0
mu0 = 1.25663706212e-6
eps0 = 1.0 / (mu0 * c0 ** 2)
Z0 = mu0 * c0


def reset_units(*args, **kwargs):
    return None
