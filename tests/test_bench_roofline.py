"""bench.py's roofline arithmetic on the CPU: the near field reports the roof that binds it (fp64
vector issue, from the counter profile of the configuration) with the HBM figure beside it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_nearfield_roof_by_vector_issue():
    import bench
    # 1.6e8 wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz) = 0.2604 ms of pure issue
    r = bench.nearfield_roof(0.39, 64.0 * 4096 * 4096, {'SQ_INSTS_VALU': 1.6e8, 'traffic_bytes': 9.1e8})
    assert r['bound'] == 'valu_fp64'
    assert abs(r['valu_issue_ms'] - 1.6e8 * 4 / (1024 * 2.4e9) * 1e3) < 1e-12
    assert abs(r['frac'] - r['valu_issue_ms'] / 0.39) < 1e-12 and 0.66 < r['frac'] < 0.68
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['hbm_frac'] - 64.0 * 4096 * 4096 / 0.39e-3 / 8e12) < 1e-12 and r['traffic'] == 9.1e8


def test_nearfield_roof_without_a_profile_falls_back_to_hbm():
    import bench
    r = bench.nearfield_roof(0.39, 64.0 * 4096 * 4096, {})
    assert r['bound'] == 'hbm' and r['valu_insts'] is None and abs(r['frac'] - r['hbm_frac']) < 1e-15


def test_pmc_key_names_the_configuration():
    import bench
    assert bench.pmc_key(1, 4096, 512, 'f64', 'auto', 1.0, 1) == \
        'gpus=1,aperture=4096,farfield=512,precision=f64,method=auto,zoom=1,pols=1'
    assert bench.pmc_key(1, 4096, 512, 'f64', 'auto', 1.0, 3).endswith('pols=3')
