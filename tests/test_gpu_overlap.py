"""The banded step (metalens_hip.h ml_step_overlap): synthesis of band b + 1 beside the row
transform of band b on a second stream.  It must be the SAME numbers as the unbanded step - the
per-sample and per-row arithmetic does not change, only which launch a sample belongs to - so the
comparison is bit for bit (to rounding where the lean row transform is used), on sizes where the bands are ragged (rows not a multiple of 8 per
band, more bands than patch rows, lenses smaller than the window).  Needs an MI355X."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KEYS = ('P', 'a_theta', 'a_phi', 'Nx', 'Ny', 'Lx', 'Ly')


def _same(got, want, exact):
    """bit for bit; with the lean row transform (which applies its stage-1 twiddles as two
    in-place products instead of one product of two tabulated values) equal to rounding"""
    for k in KEYS:
        if exact:
            assert np.array_equal(got[k], want[k], equal_nan=True), k
        else:
            assert np.array_equal(np.isnan(got[k]), np.isnan(want[k])), k
            ok = ~np.isnan(want[k])
            assert np.abs(got[k][ok] - want[k][ok]).max() <= 1e-14 * np.abs(want[k][ok]).max(), k


def _hotpath(ctx, side, M, diameter, na, world=1, rank=0):
    import bench
    from metalens_amd.pipeline import HotPath
    wl = 580e-9
    # directions = bins of the lattice of the next multiple of 256 samples (zero-padded when the
    # aperture has fewer): the grids the pruned FFT takes
    n_eff = -(-side // 256) * 256
    lens, x, u = bench.build_workload(side, M, diameter, na, wl, side / n_eff)
    src = (0.3e-6, -0.2e-6, -lens['source_distance'], 'x')
    hp = HotPath(src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'],
                 lens['hexgridset'], x, x, u, u, ctx=ctx, rank=rank, world=world)
    hp.other_source = (0.4e-6, 0.1e-6, -lens['source_distance'], 'y')
    return hp


def _fields(ctx, nx, ny):
    from metalens_amd import _lib
    F = [np.empty((nx, ny), dtype=np.complex128) for _ in range(4)]
    _lib.check(ctx.lib.ml_fields_download(ctx.handle, *[_lib.dptr(a) for a in F]))
    return F


@pytest.mark.parametrize('side,M,diameter,na', [(1024, 128, 0.25e-3, 0.5), (1000, 100, 0.2e-3, 0.3),
                                                (2048, 256, 1e-3, 0.5)])
@pytest.mark.parametrize('overlap', [(4, 4, True, 1), (7, 1, True, 2), (16, 4, False, 1), (200, 4, True, 1)])
def test_banded_step_is_bit_identical(side, M, diameter, na, overlap):
    from metalens_amd import _lib
    ctx = _lib.default_context()
    try:
        ctx.set_overlap(0)
        hp = _hotpath(ctx, side, M, diameter, na)
        hp.step()
        hp.sync()
        want = hp.results()
        assert ctx.plan_kernels()[0] == 'fft'
        F_want = _fields(ctx, side, side)
        ctx.set_overlap(*overlap)
        for _ in range(3):   # the first banded step reads the band table back; later ones only queue
            hp.step()
        hp.sync()
        got = hp.results()
        F_got = _fields(ctx, side, side)
        for a, b in zip(F_got, F_want):
            assert np.array_equal(a, b)
        _same(got, want, exact=not overlap[2])
        assert got['power_local_rows'] == want['power_local_rows']
        # another source on the same geometry: the band table is reused, the numbers follow the source
        hp.set_source(hp.other_source)
        hp.step()
        hp.sync()
        got2 = hp.results()
        ctx.set_overlap(0)
        hp.step()
        hp.sync()
        want2 = hp.results()
        _same(got2, want2, exact=not overlap[2])
        assert not np.array_equal(got2['a_theta'], got['a_theta'])
    finally:
        ctx.set_overlap(0)


@pytest.mark.parametrize('side,M,diameter,na', [(1024, 128, 0.25e-3, 0.5), (2048, 256, 1e-3, 0.5)])
@pytest.mark.parametrize('opts', [(4, True, 1), (1, False, 2), (4, True, 2)])
def test_pipelined_sweep_equals_step_by_step(side, M, diameter, na, opts):
    """ml_step_pipeline: consecutive steps overlap on two streams and two field buffers.  Whatever
    the interleaving on the GPU, what the host reads after a run of steps is the LAST step's, and
    equal to the step-by-step result (bit for bit for the fields and, with the ordinary row
    transform, for the far field) - also when the sources alternate, so that the two field buffers
    hold different fields."""
    from metalens_amd import _lib
    ctx = _lib.default_context()
    try:
        ctx.set_pipeline(False)
        hp = _hotpath(ctx, side, M, diameter, na)
        first = hp.params
        want = {}
        F_want = {}
        for name, src in (('a', None), ('b', hp.other_source)):
            if src is not None:
                hp.set_source(src)
            hp.step()
            hp.sync()
            want[name] = hp.results()
            F_want[name] = _fields(ctx, side, side)
        assert ctx.plan_kernels()[0] == 'fft'
        a_params, b_params = first, hp.params
        ctx.set_pipeline(True, *opts)
        for seq, last in (('ababab', 'b'), ('bbaba', 'a'), ('a', 'a')):
            for ch in seq:
                hp.params = a_params if ch == 'a' else b_params
                hp.step()
            hp.sync()
            got = hp.results()
            for a, b in zip(_fields(ctx, side, side), F_want[last]):
                assert np.array_equal(a, b)
            _same(got, want[last], exact=not opts[1])
            assert got['power_local_rows'] == want[last]['power_local_rows']
    finally:
        ctx.set_pipeline(False)


def test_banded_step_on_a_mirrored_shard():
    """a rank's mirrored row pairs (what bench.py --gpus 2 gives rank 0) banded = unbanded"""
    from metalens_amd import _lib
    ctx = _lib.default_context()
    try:
        ctx.set_overlap(0)
        hp = _hotpath(ctx, 1024, 128, 0.25e-3, 0.5, world=2, rank=0)
        hp.step_local()
        hp.sync()
        vec = [np.empty(hp.shape, dtype=np.complex128) for _ in range(4)]
        _lib.check(ctx.lib.ml_farfield_download(ctx.handle, *[_lib.dptr(v) for v in vec]))
        ctx.set_overlap(5, 4, True, 1)
        for _ in range(2):
            hp.step_local()
        hp.sync()
        vec2 = [np.empty(hp.shape, dtype=np.complex128) for _ in range(4)]
        _lib.check(ctx.lib.ml_farfield_download(ctx.handle, *[_lib.dptr(v) for v in vec2]))
        for a, b in zip(vec, vec2):
            assert np.abs(a - b).max() <= 1e-14 * np.abs(a).max()
    finally:
        ctx.set_overlap(0)
