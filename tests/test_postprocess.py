"""Host-side power bookkeeping (metalens_amd/postprocess.py) against the reference's own
total_P in the golden far-field fixtures and against closed forms."""
import math

import numpy as np

import golden_io
from metalens_amd import postprocess


def test_total_power_equals_reference_total_P():
    for name in ('farfield_A_lattice.npz', 'farfield_B_periphery_window.npz'):
        case = np.load(golden_io.golden_path(name))
        if case['P'].shape != (case['ux'].size, case['uy'].size):
            continue   # fixture A keeps a strided sample of P only
        got = postprocess.total_power(case['P'], float(case['dux']), float(case['duy']))
        assert abs(got - float(case['total_P'])) <= 1e-13 * abs(float(case['total_P']))


def test_encircled_power_closed_form_and_nan_handling():
    n = 801
    u = np.linspace(-1.2, 1.2, n)
    du = u[1] - u[0]
    rho2 = u[:, None] ** 2 + u[None, :] ** 2
    P = np.where(rho2 <= 1.0, 2.5, np.nan)          # uniform inside the unit circle, NaN outside
    # whole map = area of the unit disc, a cone = area of its disc (to grid resolution)
    assert abs(postprocess.total_power(P, du, du) - 2.5 * math.pi) < 0.02
    assert abs(postprocess.encircled_power(P, u, u, du, du, sin_max=0.5) - 2.5 * math.pi * 0.25) < 0.01
    assert postprocess.encircled_power(P, u, u, du, du, half_angle=math.pi / 6) == \
        postprocess.encircled_power(P, u, u, du, du, sin_max=math.sin(math.pi / 6))
    # a cone beyond the unit circle adds nothing (NaNs are skipped, not propagated)
    assert postprocess.encircled_power(P, u, u, du, du, sin_max=1.5) == postprocess.total_power(P, du, du)
    # off-axis cone
    off = postprocess.encircled_power(P, u, u, du, du, sin_max=0.2, center=(0.3, -0.4))
    assert abs(off - 2.5 * math.pi * 0.04) < 0.01
    # the curve is the same numbers in one pass
    radii = np.array([0.0, 0.1, 0.5, 0.9, 1.5])
    curve = postprocess.encircled_power_curve(P, u, u, du, du, radii)
    for r, c in zip(radii, curve):
        assert abs(c - postprocess.encircled_power(P, u, u, du, du, sin_max=r)) <= 1e-10 * max(c, 1.0)   # cumulative vs pairwise summation


def test_incoherent_sum_and_efficiency():
    u = np.linspace(-0.5, 0.5, 11)
    du = u[1] - u[0]
    a = np.full((11, 11), 1.0)
    b = np.full((11, 11), 3.0)
    b[0, 0] = np.nan
    s = postprocess.incoherent_sum([a, b, a])
    assert np.isnan(s[0, 0]) and s[5, 5] == 5.0
    total = postprocess.total_power(s, du, du)
    assert abs(total - 5.0 * 120 * du * du) < 1e-12
    assert abs(postprocess.efficiency(s, u, u, du, du, power_in=[1.0, 2.0, 1.0]) - total / 4.0) < 1e-15
    cone = postprocess.efficiency(s, u, u, du, du, [4.0], sin_max=0.15)
    assert abs(cone - 5.0 * 9 * du * du / 4.0) < 1e-12    # the 3 x 3 points with |u| <= 0.1
    assert math.isnan(postprocess.efficiency(s, u, u, du, du, power_in=0.0))
