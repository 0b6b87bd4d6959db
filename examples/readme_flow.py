"""The reference's far-field flow (README.md:27: design -> characterize -> build_nearfield -> fft2 ->
farfield_from_nearfield) on the GPU, for a synthetic 200 um lens whose tables stand in for characterize()
(S4 is out of scope).  Prints the focusing efficiency of an on-axis emitter and of one 2 um off axis.

    python examples/readme_flow.py            (needs an MI355X and the built library)

Everything named here has the reference's name and argument order; what is new is `download=False` (the near
field stays on the GPU), `farfield_from_resident_nearfield` (the reference's return tuple without the host
FFTs), `PreparedLens` (hash and upload the lens once) and `SourceSweep` (incoherent x + y + z emitters,
nearfield.py:69-73)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import metalens_amd as ma
from metalens_amd import layout, synthetic
from metalens_amd.constants import nm, um


def main(radius=100 * um, numerical_aperture=0.5, wavelength=580 * nm, verbose=True):
    lens = synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet), layout.make_design,
                               radius=radius, numerical_aperture=numerical_aperture, wavelength=wavelength,
                               periphery_orders='physical')
    f = lens['source_distance']
    periphery, centre, hgs = lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset']
    # --- one source, the reference's two calls (its default grid: good_fft_number(2 r / (lambda / 2.2)) samples)
    _, _, _, _, x, y, power_in, n_glass = ma.build_nearfield(
        source_x=0, source_y=0, source_z=-f, source_pol='x', wavelength=wavelength,
        lens_periphery_summary=periphery, lens_center_summary=centre, hexgridset=hgs, download=False)
    P, total_P, ux, uy, dux, duy = ma.farfield_from_resident_nearfield(x, y, wavelength, n_glass)
    # a collimator: the power within 1 degree of the axis over the power through the lens
    cone = np.hypot(ux, uy) <= math.sin(math.radians(1.0))
    in_cone = np.nansum(np.where(cone, P, 0.0)) * dux * duy
    if verbose:
        print('%d x %d aperture samples; power through the lens %.4g W, far field %.4g W (%.1f %%), within 1 degree %.1f %%'
              % (len(x), len(y), power_in, total_P, 100 * total_P / power_in, 100 * in_cone / power_in))
    # --- an unpolarised emitter 2 um off axis: x + y + z dipoles summed incoherently on the GPU
    u = ma.fft_direction_cosines(len(x), x[1] - x[0], wavelength, n_glass)
    u = np.sort(u[np.abs(u) <= 0.1])                      # the central window of the lattice
    sweep = ma.SourceSweep(wavelength, periphery, centre, hgs, x, y, u, u)
    res = sweep.run([(2 * um, 0.0, -f, pol) for pol in 'xyz'], cone=math.sin(math.radians(2.0)))
    if verbose:
        print('emitter 2 um off axis, x + y + z: %.1f %% of the power through the lens arrives in |u| <= 0.1, '
              '%.1f %% within 2 degrees of the axis' % (100 * res['efficiency'], 100 * res['cone_efficiency']))
    return {'power_in': power_in, 'total_P': total_P, 'in_cone': in_cone, 'efficiency_off_axis': res['efficiency'],
            'cone_efficiency_off_axis': res['cone_efficiency']}


if __name__ == '__main__':
    main()
