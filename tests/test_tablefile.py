"""Binary table files: round trip of characterisation records and packed
grids; a loaded object gives bit-identical grids when its interpolators are rebuilt."""
import math

import numpy as np

from metalens_amd import synthetic, tablefile
from metalens_amd.grating import Grating, GratingCollection
from metalens_amd.lens_center import HexGridSet

nm, degree = 1e-9, math.pi / 180


def test_collection_round_trip(tmp_path):
    gc = synthetic.make_collection(Grating, GratingCollection, 20 * degree, 28 * degree, 580 * nm,
                                   num_gratings=6, drop_every=5)
    gc.build_interpolators()
    p = str(tmp_path / 'gc.npz')
    tablefile.save(p, gc)
    back = tablefile.load(p)
    assert back.lens_type == 'round' and back.target_wavelength == gc.target_wavelength
    assert [g.grating_period for g in back.grating_list] == [g.grating_period for g in gc.grating_list]
    assert back.grating_list[2].data == gc.grating_list[2].data
    assert back.interpolator_bounds == gc.interpolator_bounds
    for k, f in gc.interpolators.items():
        assert np.array_equal(back.interpolators[k].values, f.values)
        assert all(np.array_equal(a, b) for a, b in zip(back.interpolators[k].grid, f.grid))
    # rebuilding from the loaded records reproduces the packed grids bit for bit
    stored = {k: v.values.copy() for k, v in back.interpolators.items()}
    back.build_interpolators()
    for k, v in stored.items():
        assert np.array_equal(back.interpolators[k].values, v)


def test_hexgridset_round_trip(tmp_path):
    hgs = synthetic.make_hexgridset(Grating, HexGridSet, 580 * nm, num_entries=5)
    hgs.build_interpolators()
    p = str(tmp_path / 'hgs.npz')
    tablefile.save(p, hgs)
    back = tablefile.load(p)
    assert np.array_equal(back.x_amp_list, hgs.x_amp_list)
    assert back.sep == hgs.sep and len(back.grating_list) == 5
    assert set(back.interpolators) == set(hgs.interpolators)
    for k, f in hgs.interpolators.items():
        assert np.array_equal(back.interpolators[k].values, f.values)
    assert back.pick_from_phase(1.0) == hgs.pick_from_phase(1.0)
