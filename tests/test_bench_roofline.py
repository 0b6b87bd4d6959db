"""bench.py's roofline arithmetic on the CPU: the near field reports the roof that binds it (fp64
vector issue, from the counter profile of the configuration) with the HBM figure beside it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_nearfield_roof_is_the_hbm_figure_with_the_issue_diagnosis_beside_it():
    import bench
    # the contract's object: algorithmic bytes (64 B per sample) / launch time against 8 TB/s
    r = bench.nearfield_roof(0.39, 64.0 * 4096 * 4096, {'SQ_INSTS_VALU': 1.6e8, 'traffic_bytes': 9.1e8})
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['achieved'] - 64.0 * 4096 * 4096 / 0.39e-3 / 1e9) < 1e-9
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-15 and r['traffic'] == 9.1e8
    # ... and what binds: 1.6e8 wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz) = 0.2604 ms of pure issue
    v = r['valu']
    assert abs(v['issue_ms'] - 1.6e8 * 4 / (1024 * 2.4e9) * 1e3) < 1e-12
    assert abs(v['issue_frac'] - v['issue_ms'] / 0.39) < 1e-12 and 0.66 < v['issue_frac'] < 0.68
    assert v['stale'] is False and bench.nearfield_roof(0.39, 1.0, {'SQ_INSTS_VALU': 1.0}, stale=True)['valu']['stale']


def test_nearfield_roof_without_a_profile_has_no_counters():
    import bench
    r = bench.nearfield_roof(0.39, 64.0 * 4096 * 4096, {})
    assert r['bound'] == 'hbm' and r['traffic'] is None and 'valu' not in r


def test_kernel_source_id_is_stable():
    import bench
    a = bench.kernel_source_id()
    assert a == bench.kernel_source_id() and len(a) == 16


def test_pmc_key_names_the_configuration():
    import bench
    assert bench.pmc_key(1, 4096, 512, 'f64', 'auto', 1.0, 1) == \
        'gpus=1,aperture=4096,farfield=512,precision=f64,method=auto,zoom=1,pols=1'
    assert bench.pmc_key(1, 4096, 512, 'f64', 'auto', 1.0, 3).endswith('pols=3')
