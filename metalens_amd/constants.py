"""Physical constants and length units used on the near-field / far-field path.

The reference takes ``c0`` and ``Z0`` from the un-vendored ``numericalunits``
package (reference nearfield.py:213,221-222,227-228,308,422;
nearfield_farfield.py:183).  That package is absent here and its values have
changed between releases, so every entry point of this package
takes ``c0``/``Z0`` as explicit keyword arguments that default to the values
below, and every golden fixture records the values it was generated with.

The module-level names are SI: metre = second = coulomb = 1.  A caller whose lengths, ``c0`` and
``Z0`` live in ANOTHER unit system - the reference's users do: ``numericalunits`` draws a random
value for every base unit per process and the reference never resets it (reference
nearfield.py:14-15) - passes ``units=`` to the entry points: any object with the attributes ``nm``,
``c0`` and ``Z0`` (and, for the default dipole moment, ``C`` and ``m``), e.g. the ``numericalunits``
module itself.  ``Units`` below builds one; ``as_units`` is what the entry points call.
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


class Units:
    """A unit system in the sense of ``numericalunits``: the value of every base unit, everything
    else derived the way that package derives it.  ``Units()`` is SI (= this module's names)."""

    def __init__(self, m=1.0, kg=1.0, s=1.0, C=1.0):
        self.m, self.kg, self.s, self.C = m, kg, s, C
        self.um, self.nm = 1e-6 * m, 1e-9 * m
        self.V = kg * m ** 2 / (s ** 2 * C)
        self.c0 = c0 * m / s
        self.mu0 = mu0 * kg * m / C ** 2
        self.eps0 = 1.0 / (self.mu0 * self.c0 ** 2)
        self.Z0 = self.mu0 * self.c0


SI = Units()


def as_units(units):
    """``None`` -> SI; anything exposing ``nm``, ``c0`` and ``Z0`` is taken as it is"""
    if units is None:
        return SI
    for name in ('nm', 'c0', 'Z0'):
        if not hasattr(units, name):
            raise TypeError('units= needs the attributes nm, c0 and Z0 (e.g. the numericalunits module); '
                            '%r has no %s' % (units, name))
    return units


def default_dipole_moment(units):
    """the reference's default, 1e-30 C m (nearfield.py:68), in the caller's units"""
    return 1e-30 * getattr(units, 'C', 1.0) * getattr(units, 'm', 1e9 * units.nm)
