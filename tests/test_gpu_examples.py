"""examples/ run as written (GPU): the reference's README flow through this package"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_readme_flow_example_runs_and_conserves_power():
    spec = importlib.util.spec_from_file_location('readme_flow', os.path.join(ROOT, 'examples', 'readme_flow.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(radius=50e-6, verbose=False)
    # (the tables are synthetic - smooth made-up amplitudes, not a physical grating - so the only laws are
    # positivity and that a part is no more than the whole)
    import math
    assert all(math.isfinite(v) for v in out.values())
    assert out['power_in'] > 0 and out['total_P'] > 0
    assert 0 <= out['in_cone'] <= out['total_P'] * (1 + 1e-12)
    assert out['efficiency_off_axis'] > 0
    assert 0 <= out['cone_efficiency_off_axis'] <= out['efficiency_off_axis'] * (1 + 1e-12)
