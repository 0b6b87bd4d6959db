"""The whole-flow tests of the HIP path: every test here starts `bench.py` (or ranks of it) as a
SUBPROCESS on the GPU - real RCCL with one rank, 2 / 4 / 8 ranks sharing one GPU through the file
communicator, the wavelength replicas, the bench-line contract, real RCCL with two ranks where two
devices exist.  Kept apart from tests/test_gpu_parity.py and named to run LAST: under `pytest -x` a
flaky subprocess must not hide the kernel-parity tests from the record.  Needs an MI355X: ``-m gpu``."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('reduce', ['amplitudes', 'vectors'])
def test_rccl_path_single_rank(reduce):
    """the multi-GPU code path (RCCL loaded with dlopen, unique-id exchange through /tmp,
    communicator, all-reduce of the projected amplitudes or of the radiation vectors, max/sum
    reductions) run for real with one rank: results must equal the plain single-GPU run"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--aperture', '512', '--farfield', '64',
           '--diameter', '3e-4', '--steps', '2', '--warmup', '1', '--blocks', '1',
           '--cpu-rows', '0', '--cpu-fft-side', '0', '--reduce', reduce]
    env = dict(os.environ, ML_FORCE_RCCL='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               MASTER_ADDR='127.0.0.1', MASTER_PORT='29511' if reduce == 'vectors' else '29512')
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # stdout must be the ONE JSON line and nothing else (RCCL's version banner, which the library
    # prints to stdout when the first communicator is created, is diverted to stderr)
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1
    assert line['rel_err']['farfield_E_vs_oracle'] < 1e-12
    assert line['rel_err']['nearfield_vs_oracle'] < 1e-12


def _run_bench(extra, env, timeout=600, aperture=512, farfield=64, diameter='3e-4'):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--aperture', str(aperture),
           '--farfield', str(farfield),
           '--diameter', diameter, '--na', '0.5', '--steps', '2', '--warmup', '1', '--blocks', '1',
           '--cpu-rows', '0', '--cpu-fft-side', '0', '--scaling', 'strong'] + extra
    return subprocess.Popen(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize('reduce,aperture,pairs,world', [
    ('amplitudes', 512, 0, 2), ('vectors', 512, 0, 2), ('amplitudes', 511, 0, 2), ('amplitudes', 512, 300, 2),
    ('vectors', 512, 300, 2), ('amplitudes', 2048, 0, 4), ('amplitudes', 2048, 0, 8), ('amplitudes', 1000, 0, 4),
    ('amplitudes-allreduce', 512, 0, 2), ('amplitudes-allreduce', 2048, 0, 4), ('amplitudes', 512, 301, 2)])
def test_two_ranks_sharing_one_gpu(tmp_path, reduce, aperture, pairs, world):
    """bench.py --gpus N end to end on ONE GPU: N processes (ranks 0 .. N-1, all on device 0)
    with the test communicator (ML_COMM_BACKEND=file; RCCL refuses two ranks per GPU): unique-id
    rendezvous, row shards (interleaved blocks on lattice grids - 512 and 2048 rows -, weighted
    mirrored pairs at 1000 rows, whose lattice is not a multiple of 256 N), per-rank synthesis and
    transform, the reduction - a reduce-scatter over blocks of direction rows + each rank's power of its
    block + results()' all-gather by default; the all-reduce forms; the all-reduce fallback when the
    directions do not divide by the rank count (301 listed directions over 2 ranks) - max-over-ranks
    timing - and the far field must equal the one-process result.  The odd aperture takes contiguous row blocks, and only the rank that owns the x = 0
    row meets nearest-cell ties: results() has to settle them collectively.  ``pairs`` > 0: a LIST
    of directions instead of the tensor grid (no folded / mirrored form: contiguous row blocks)."""
    import json
    more = ['--pair-list', str(pairs)] if pairs else []
    one = str(tmp_path / 'one.npz')
    p = _run_bench(['--dump', one] + more, {}, aperture=aperture)
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, err[-2000:]
    two = str(tmp_path / 'two.npz')
    env = dict(ML_COMM_BACKEND='file', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(29533 + ('amplitudes', 'vectors', 'amplitudes-allreduce').index(reduce) + 3 * (aperture % 2) +
                               6 * (pairs > 0) + 12 * (pairs % 2) + 24 * world))
    procs = [_run_bench(['--gpus', str(world), '--dump', two, '--reduce', reduce] + more,
                        dict(env, RANK=str(r), LOCAL_RANK=str(r)), aperture=aperture)
             for r in range(world)]
    outs = [q.communicate(timeout=600) for q in procs]
    for q, (o, e) in zip(procs, outs):
        assert q.returncode == 0, e[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1 and not outs[1][0].strip(), (outs[0][0], outs[1][0])
    line = json.loads(lines[0])
    assert line['n_gpus'] == world and line['scaling'] == 'strong'
    # the diagnostics a first real multi-GPU run is read by: the backend's own rank count, every rank's
    # kernel times, what the main stream waited for the collective, the same shards without it
    mg = line['multi_gpu']
    assert mg['ranks_reported_by_backend'] == world and mg['backend'].startswith('file') and mg['reduce'] == reduce
    assert [r['rank'] for r in mg['per_rank_ms']] == list(range(world))
    assert all(r['nearfield'] > 0 and r['zgemm_stage1'] > 0 for r in mg['per_rank_ms'])
    assert mg['ms_per_step_no_collective'] > 0
    want_sharding = ('interleaved' if not pairs and aperture % (256 * world) == 0 else
                     'mirrored' if not pairs and aperture % 2 == 0 else 'rows')
    assert line['config']['sharding'].startswith(want_sharding), line['config']['sharding']
    a, b = np.load(one), np.load(two)
    for key in ('a_theta', 'a_phi'):
        assert np.abs(a[key] - b[key]).max() <= 1e-13 * np.abs(a[key]).max(), key
    ok = ~np.isnan(a['P'])
    assert np.array_equal(np.isnan(b['P']), ~ok)
    assert np.abs(a['P'][ok] - b['P'][ok]).max() <= 1e-12 * a['P'][ok].max()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_plain_bench_gpus_n_starts_its_own_ranks(tmp_path, world):
    """``python bench.py --gpus N`` with NO launcher environment (the way the driver starts the N = 1
    line): bench.py starts the N ranks itself and rank 0's stdout carries the one JSON line.  Here the
    ranks share the one GPU through the file communicator; the far field equals the one-process one."""
    import json
    one = str(tmp_path / 'one.npz')
    p = _run_bench(['--dump', one], {}, aperture=2048)
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, err[-2000:]
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    many = str(tmp_path / 'many.npz')
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--aperture', '2048', '--farfield', '64',
           '--diameter', '3e-4', '--na', '0.5', '--steps', '2', '--warmup', '1', '--blocks', '1', '--cpu-rows', '0',
           '--cpu-fft-side', '0', '--scaling', 'strong', '--dump', many]
    q = subprocess.run(cmd, env=dict(env, ML_COMM_BACKEND='file'), capture_output=True, text=True, timeout=900)
    assert q.returncode == 0, q.stderr[-2000:]
    lines = [l for l in q.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, q.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == world and line['multi_gpu']['ranks_reported_by_backend'] == world
    assert line['config']['sharding'].startswith('interleaved')
    # the line carries the SAME workload on one GPU (rank 0 alone on the whole aperture): the N = 1 point
    # of the strong-scaling curve, whatever the driver ran at --gpus 1
    ref = line['multi_gpu']['one_gpu_same_workload']
    assert ref['ms_per_step'] > 0 and abs(ref['value'] - 2048.0 ** 2 * 64 ** 2 / (ref['ms_per_step'] * 1e-3)) < 1e-6 * ref['value']
    a, b = np.load(one), np.load(many)
    for key in ('a_theta', 'a_phi'):
        assert np.abs(a[key] - b[key]).max() <= 1e-13 * np.abs(a[key]).max(), key


def test_wavelength_replicas(tmp_path):
    """BASELINE configs[3] (tri-wavelength sweep, one wavelength per rank, no collective in the data
    path): ``bench.py --replicas wavelength`` as three single-rank runs, one per wavelength, then as
    two ranks sharing the GPU through the file communicator.  Every replica checks itself against the
    oracle and reports the substrate index it ran with - explicit, because none of the three
    wavelengths is in the reference's table (grating.py:1277-1288, nearfield.py:111-113)."""
    import json
    want = {450: 1.4656, 532: 1.4607, 635: 1.4570}
    extra = ['--replicas', 'wavelength', '--check', '1', '--cold', '0']
    for k, (nm_, ng) in enumerate(want.items()):
        p = _run_bench(extra + ['--replica-index', str(k)], {}, aperture=512)
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        d = json.loads([l for l in out.splitlines() if l.strip()][-1])
        (rep,) = d['config']['replicas']
        assert round(rep['wavelength_nm']) == nm_ and rep['n_glass'] == ng
        assert 0 <= d['rel_err']['nearfield_vs_oracle'] < 1e-12 and 0 <= d['rel_err']['farfield_E_vs_oracle'] < 1e-12
        assert d['config']['parallelism'].startswith('replicas only')
    # one wavelength at the north-star aperture (4096^2 -> 512^2 on the 1 mm lens)
    p = _run_bench(extra + ['--replica-index', '1'], {}, aperture=4096, farfield=512, diameter='1e-3')
    out, err = p.communicate(timeout=900)
    assert p.returncode == 0, err[-2000:]
    d = json.loads([l for l in out.splitlines() if l.strip()][-1])
    (rep,) = d['config']['replicas']
    assert round(rep['wavelength_nm']) == 532 and rep['n_glass'] == 1.4607 and d['config']['aperture'] == 4096
    assert 0 <= d['rel_err']['nearfield_vs_oracle'] < 1e-12 and 0 <= d['rel_err']['farfield_E_vs_oracle'] < 1e-12
    env = dict(ML_COMM_BACKEND='file', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29611')
    procs = [_run_bench(['--gpus', '2'] + extra, dict(env, RANK=str(r), LOCAL_RANK=str(r)), aperture=512)
             for r in range(2)]
    outs = [q.communicate(timeout=600) for q in procs]
    for q, (o, e) in zip(procs, outs):
        assert q.returncode == 0, e[-2000:]
    d = json.loads([l for l in outs[0][0].splitlines() if l.strip()][-1])
    assert d['n_gpus'] == 2 and [round(r['wavelength_nm']) for r in d['config']['replicas']] == [450, 532]
    assert [r['n_glass'] for r in d['config']['replicas']] == [1.4656, 1.4607]
    for r in d['config']['replicas']:
        assert 0 <= r['nearfield_vs_oracle'] < 1e-12 and 0 <= r['farfield_E_vs_oracle'] < 1e-12
    # twice the work of one replica in the same time: the aggregate counts both apertures
    assert abs(d['value'] - 2 * 512.0 ** 2 * 64 ** 2 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']


def test_bench_line_contract():
    """bench.py at N = 1 prints exactly one JSON line with the fields the driver reads: the
    metric, K timed steps, a roofline object with a fraction <= 1 and the per-launch duration it
    came from, the CPU baseline timed beside it, and the self-check against the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--aperture', '512', '--farfield', '64',
           '--diameter', '1.2e-4', '--steps', '3', '--warmup', '1', '--blocks', '2', '--cpu-rows', '32',
           '--cpu-fft-side', '128']
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['metric'] == 'aperture x far-field pair-evals/sec' and d['unit'] == 'pair-evals/s'
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert abs(d['value'] - 512.0 ** 2 * 64 ** 2 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
    assert len(d['ms_per_step_blocks']) == 2 and 'workload' in d['config']
    for key in ('roofline', 'roofline_other'):
        r = d[key]
        # the contract's object: algorithmic bytes (or executed flops) per launch / launch time against the peak
        assert r['bound'] in ('hbm', 'mfma') and 0 < r['frac'] <= 1 and r['peak'] > 0
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['avg_launch_ms'] > 0 and 'traffic' in r
        if r['kernel'].startswith('nearfield'):
            # 64 B per sample the launches process - the samples inside the lens circle - written once; the figure
            # over the whole window beside it; what binds the kernel (vector-instruction issue) rides along where
            # the configuration has a counter profile
            n_in = d['config']['samples_in_lens']
            assert 0.5 * 512 * 512 < n_in < 512 * 512       # (the 120 um lens in its 512^2 window at pitch lambda/2.2)
            assert r['bound'] == 'hbm' and abs(r['bytes_per_launch'] - 64.0 * n_in) < 1
            assert abs(r['frac_full_grid'] - r['frac'] * 512 * 512 / n_in) < 1e-9
            assert 'valu' not in r or 0 < r['valu']['issue_frac'] <= 1
        else:
            assert 'traffic_frac' in r
    assert 0 < d['roofline']['step_hbm_frac'] <= 1
    assert d['config']['pmc_key'].startswith('gpus=1,aperture=512,farfield=64,precision=f64')
    # the tables of the timed workload hold the three orders SURVEY.md 8(d) prescribes; the line also carries the
    # same workload with the order lists characterize() would record, and says which kernels each took
    assert d['config']['orders'] == 'survey' and d['config']['orders_per_table'] == [3] * len(d['config']['orders_per_table'])
    assert d['config']['nearfield_kernels']['family'] == 'orders-along-x'
    ph = d['physical_orders']
    assert max(ph['orders_per_table']) > 4 and ph['nearfield_kernels']['ring_orders_max'] == max(ph['orders_per_table'])
    assert ph['ms_per_step'] > 0 and ph['nearfield_ms'] > 0
    # a single call on a grid the context has not seen (geometry kernel, scans, zeros stored)
    assert d['ms_first_step_new_geometry'] > 0 and d['cold_step']['first_ms'] >= d['cold_step']['third_ms'] > 0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0 and 'sample' in c
    assert d['cpu_baseline_reference_route']['value'] > 0
    assert d['rel_err']['nearfield_vs_oracle'] < 1e-12 and d['rel_err']['farfield_E_vs_oracle'] < 1e-12


def test_two_gpus_real_rccl(tmp_path):
    """bench.py --gpus 2 with REAL RCCL over xGMI, one rank per GPU: runs only where two GPUs are
    visible (the single-GPU boxes of this pool skip it; the file-communicator test above covers
    the same flow there).  Rendezvous through the launcher's environment, ncclCommInitRank with
    two ranks, the all-reduce of the projected amplitudes on the second stream, max-over-ranks
    timing - and the far field must equal the one-process result."""
    import json
    from metalens_amd import _lib
    n = _lib.c_int(0)
    _lib.check(_lib.load().ml_device_count(_lib.byref(n)))
    if n.value < 2:
        pytest.skip('needs two GPUs (%d visible)' % n.value)
    one = str(tmp_path / 'one.npz')
    p = _run_bench(['--dump', one, '--steps', '6'], {}, aperture=512)
    out, err = p.communicate(timeout=600)
    assert p.returncode == 0, err[-2000:]
    for reduce in ('amplitudes', 'vectors'):
        two = str(tmp_path / ('two_%s.npz' % reduce))
        env = dict(WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(29571 + (reduce == 'vectors')), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs = [_run_bench(['--gpus', '2', '--dump', two, '--reduce', reduce, '--steps', '6'],
                            dict(env, RANK=str(r), LOCAL_RANK=str(r)), aperture=512) for r in range(2)]
        outs = [q.communicate(timeout=600) for q in procs]
        for q, (o, e) in zip(procs, outs):
            assert q.returncode == 0, e[-2000:]
        line = json.loads([l for l in outs[0][0].splitlines() if l.strip()][-1])
        assert line['n_gpus'] == 2
        # what RCCL itself says it connected (ncclCommCount), not what the launcher asked for
        assert line['multi_gpu']['backend'] == 'rccl' and line['multi_gpu']['ranks_reported_by_backend'] == 2, line['multi_gpu']
        a, b = np.load(one), np.load(two)
        for key in ('a_theta', 'a_phi'):
            assert np.abs(a[key] - b[key]).max() <= 1e-13 * np.abs(a[key]).max(), (reduce, key)
        ok = ~np.isnan(a['P'])
        assert np.array_equal(np.isnan(b['P']), ~ok)
        assert np.abs(a['P'][ok] - b['P'][ok]).max() <= 1e-12 * a['P'][ok].max()
