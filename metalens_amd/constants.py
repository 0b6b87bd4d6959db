"""Physical constants and length units used on the near-field / far-field path.

The reference takes ``c0`` and ``Z0`` from the un-vendored ``numericalunits``
package (reference nearfield.py:213,221-222,227-228,308,422;
nearfield_farfield.py:183).  That package is absent here and its values have
changed between releases, so every entry point of this package
takes ``c0``/``Z0`` as explicit keyword arguments that default to the values
below, and every golden fixture records the values it was generated with.

SI throughout: metre = second = coulomb = 1.
"""
from math import pi

m = 1.0
um = 1e-6
nm = 1e-9
C = 1.0
V = 1.0

#: speed of light in vacuum (exact, SI 2019)
c0 = 299792458.0
#: vacuum permeability, CODATA 2018
mu0 = 1.25663706212e-6
#: vacuum permittivity derived from the two above
eps0 = 1.0 / (mu0 * c0 ** 2)
#: impedance of free space derived from the two above
Z0 = mu0 * c0

degree = pi / 180
inf = float('inf')
