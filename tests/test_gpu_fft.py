"""The output-pruned FFT path of the aperture -> direction transform (csrc/zfft.hip): taken when a
direction grid is a run of consecutive bins of the aperture's FFT lattice (the reference's own
far-field grid, nearfield_farfield.py:35-39).  Checked against the CPU oracle's direct sum, against
the GEMM path on the same inputs, and through the sharded entry points.  Needs an MI355X."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12
WL, N_GLASS = 580e-9, 1.459


@pytest.fixture(scope='module')
def ma():
    import metalens_amd
    return metalens_amd


@pytest.fixture(params=['auto', 'fft-streamed'])
def ctx(request):
    """both layouts of the stage-1 result: row-major (what 'auto' takes at these sizes) and transposed
    for a streaming stage 2 ('fft-streamed': what 'auto' takes from 96 MiB on) -
    whole apertures, contiguous and mirrored row shards, accumulation"""
    from metalens_amd import _lib
    c = _lib.default_context()
    c.set_method(request.param)
    c.method_name = request.param
    c.set_precision('f64')
    yield c
    c.set_method('auto')


def lattice(n_eff, step, m, j0):
    """m consecutive bins, starting at bin j0, of the FFT lattice of n_eff samples `step` apart"""
    return (np.arange(m) + j0) * ((WL / N_GLASS) / (step * n_eff))


def fields(nx, ny, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((nx, ny)) + 1j * rng.standard_normal((nx, ny)) for _ in range(4)]


@pytest.mark.parametrize('nx,ny,nex,ney,mx,my,jx,jy', [
    (256, 512, 256, 512, 64, 100, -32, -50),        # R3 = 1 and 2, windows around the axis
    (256, 512, 256, 512, 256, 512, -128, -256),     # every bin: the whole FFT
    (1024, 1280, 1024, 1280, 128, 77, -64, -3),     # R3 = 4 and 5 (a non-power-of-two residue count)
    (700, 2048, 768, 2048, 96, 256, -48, -128),     # x zero-padded to 3 * 256, y: R3 = 8
    (64, 4096, 256, 4096, 33, 512, -16, -256),      # the benchmark's y axis (R3 = 16)
    (48, 8192, 512, 8192, 40, 512, 7, 3000),        # R3 = 32, windows far off axis
    (300, 1000, 1024, 2048, 64, 64, -40, -20),      # finer than the lattice: zoom 1/3.4 and 1/2
    (40, 4000, 256, 4096, 16, 300, -8, 3990),       # window wrapping around the end of the lattice
    (24, 16384, 256, 16384, 16, 600, -8, -300),     # beyond one workgroup's LDS: two interleaved halves of 8192
    (20, 12288, 512, 12288, 24, 100, -12, -50),     # R3 = 48 -> two sub-sequences of R3 = 24
    (16, 20000, 256, 24576, 16, 90, -8, -45),       # R3 = 96 -> three of R3 = 32, zero-padded axis
    (16384, 12, 16384, 256, 70, 12, -35, -6),       # ... and along x (the column pass)
    # lattices that are not multiples of 256 long (what good_fft_number hands out, nearfield.py:30-36):
    (400, 1920, 400, 1920, 400, 256, -200, -128),   # 16 x 25 and 128 x 15: on the 16 / 2 times finer lattice
    (960, 1152, 960, 1152, 100, 1152, -50, -576),   # 64 x 15, 128 x 9
    (2000, 48, 2000, 300, 64, 20, -32, -10),        # 16 x 125: five sub-sequences of 6400; 4 x 75 -> 64 x 300
    (90, 3600, 256, 3600, 16, 128, -8, -64),        # 16 x 225: nine sub-sequences
])
def test_lattice_grids_take_the_fft_and_match_the_oracle(ma, ctx, nx, ny, nex, ney, mx, my, jx, jy):
    from oracle import farfield_oracle
    px, py = WL / 2.2, WL / 2.3
    x = (np.arange(nx) - 3.3) * px
    y = (np.arange(ny) + 11.1) * py
    # the lattice is defined by the spacing the axis arrays actually have (as the reference
    # does, nearfield_farfield.py:22-23,35-36), which differs from the nominal pitch by ~1e-13
    ux, uy = lattice(nex, x[1] - x[0], mx, jx), lattice(ney, y[1] - y[0], my, jy)
    F = fields(nx, ny, nx + ny)
    got = ma.farfield_direct(*F, x, y, WL, N_GLASS, ux, uy, ctx=ctx)
    # ('auto' takes the FFT on lattices padded at most four-fold and leaves the others to the GEMMs, which are faster
    # there - metalens_hip.h ML_METHOD_AUTO; 'fft-streamed' takes it wherever there is one)
    forced = ctx.method_name == 'fft-streamed'
    for axis, (kernel, n_lattice) in enumerate(zip(ctx.plan_kernels(), (ney, nex))):   # (stage 1 = y, stage 2 = x)
        assert (kernel == 'fft') == (forced or 256 // math.gcd(n_lattice, 256) <= 4), (axis, kernel)
    want = farfield_oracle.farfield_direct(*F, x, y, WL, N_GLASS, ux, uy)
    for key in ('Nx', 'Ny', 'Lx', 'Ly'):
        assert np.abs(got[key] - want[key]).max() <= TOL * np.abs(want[key]).max(), key
    ok = ~np.isnan(want['P'])                 # directions inside the unit circle
    assert np.array_equal(np.isnan(got['P']), ~ok)
    if ok.any():
        for key in ('a_theta', 'a_phi'):
            assert np.abs(got[key][ok] - want[key][ok]).max() <= TOL * np.abs(want[key][ok]).max(), key
        assert np.abs(got['P'][ok] - want['P'][ok]).max() <= 1e-11 * want['P'][ok].max()


def test_one_axis_on_the_lattice_the_other_off(ma, ctx):
    """each axis decides for itself: FFT where the grid sits on the lattice, GEMM elsewhere"""
    from oracle import farfield_oracle
    nx, ny, mx, my = 512, 768, 50, 60
    p = WL / 2.2
    x, y = np.arange(nx) * p, np.arange(ny) * p
    F = fields(nx, ny, 4)
    on_x, on_y = lattice(nx, x[1] - x[0], mx, -25), lattice(ny, y[1] - y[0], my, -30)
    off_x, off_y = np.linspace(-0.31, 0.3, mx), np.linspace(-0.2, 0.22, my)
    for ux, uy, kernels in ((on_x, off_y, ('folded', 'fft')), (off_x, on_y, ('fft', 'folded')),
                            (on_x * (1 + 1e-7), on_y, ('fft', 'folded'))):
        got = ma.farfield_direct(*F, x, y, WL, N_GLASS, ux, uy, ctx=ctx)
        assert ctx.plan_kernels()[0] == kernels[0] and (ctx.plan_kernels()[1] == 'fft') == (kernels[1] == 'fft')
        want = farfield_oracle.farfield_direct(*F, x, y, WL, N_GLASS, ux, uy)
        for key in ('Nx', 'Ny', 'Lx', 'Ly'):
            assert np.abs(got[key] - want[key]).max() <= TOL * np.abs(want[key]).max(), key


@pytest.mark.parametrize('n,m', [(1024, 256), (2048, 256)])
def test_fft_equals_gemm_on_the_same_fields(ma, ctx, n, m):
    """the two formulations of one sum: same radiation vectors, amplitudes and power"""
    p = WL / 2.2
    x = (np.arange(n) - (n - 1) / 2) * p
    u = lattice(n, x[1] - x[0], m, -(m // 2))
    F = fields(n, n, n)
    a = ma.farfield_direct(*F, x, x, WL, N_GLASS, u, u, ctx=ctx)
    assert ctx.plan_kernels() == ('fft', 'fft')
    ctx.set_method('gemm')
    b = ma.farfield_direct(*F, x, x, WL, N_GLASS, u, u, ctx=ctx)
    assert ctx.plan_kernels() == ('folded', 'folded')
    for key in ('Nx', 'Ny', 'Lx', 'Ly', 'a_theta', 'a_phi', 'P'):
        assert np.abs(a[key] - b[key]).max() <= 1e-13 * np.abs(b[key]).max(), key


@pytest.mark.parametrize('mirrored', [False, True])
def test_sharded_rows_through_the_fft(ma, ctx, mirrored):
    """row shards (contiguous blocks or mirrored pairs, as the ranks of a multi-GPU run own
    them) accumulate to the whole aperture: the column pass treats rows it does not hold as zero"""
    from metalens_amd import _lib, dist
    nx, ny, mx, my = 512, 256, 96, 64
    p = WL / 2.2
    x, y = np.arange(nx) * p, np.arange(ny) * p
    ux, uy = lattice(nx, x[1] - x[0], mx, -48), lattice(ny, y[1] - y[0], my, -32)
    F = fields(nx, ny, 9)
    whole = ma.farfield_direct(*F, x, y, WL, N_GLASS, ux, uy, ctx=ctx)
    t = ma.FarfieldTransform(nx, ny, x[1] - x[0], y[1] - y[0], WL, N_GLASS, ux, uy, ctx=ctx)
    assert ctx.plan_kernels() == ('fft', 'fft')
    world = 3
    for rank in range(world):
        if mirrored:
            q0, q1 = dist.mirrored_block(nx, world, rank, align=2)
            rows = dist.mirrored_rows(nx, q0, q1)
        else:
            q0, q1 = dist.row_block(nx, world, rank)
            rows = np.arange(q0, q1)
        part = [np.ascontiguousarray(f[rows]) for f in F]
        _lib.check(ctx.lib.ml_fields_upload(ctx.handle, len(rows), ny, *[_lib.dptr(a) for a in part]))
        t.transform(row0=q0, accumulate=rank > 0, mirrored=mirrored)
    got = t.radiation_vectors()
    for key in ('Nx', 'Ny', 'Lx', 'Ly'):
        assert np.abs(got[key] - whole[key]).max() <= 1e-13 * np.abs(whole[key]).max(), key


def test_hot_path_fft_equals_gemm(ma, ctx):
    """the resident pipeline (synthesis -> transform -> projection) on the benchmark's grid:
    FFT and GEMM formulations agree, and the synthesis does not pre-modulate for the FFT"""
    import math
    from metalens_amd import layout, synthetic
    from metalens_amd.pipeline import HotPath
    lens = synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet),
                               layout.make_design, radius=0.14e-3, numerical_aperture=0.5,
                               wavelength=WL, switch_angle=12 * math.pi / 180, num_gratings=20,
                               num_entries=12)
    n, m = 1024, 128
    p = WL / 2.2
    x = (np.arange(n) - (n - 1) / 2) * p
    u = lattice(n, x[1] - x[0], m, -(m // 2))
    src = (0.4e-6, -0.3e-6, -lens['source_distance'], 'y')
    out = {}
    for method in ('auto', 'gemm'):
        hp = HotPath(src, WL, lens['lens_periphery_summary'], lens['lens_center_summary'],
                     lens['hexgridset'], x, x, u, u, ctx=ctx, method=method)
        hp.step()
        hp.sync()
        out[method] = hp.results()
        assert ctx.plan_kernels() == (('fft', 'fft') if method == 'auto' else ('folded', 'folded'))
    for key in ('Nx', 'Ny', 'Lx', 'Ly', 'a_theta', 'a_phi', 'P'):
        a, b = out['auto'][key], out['gemm'][key]
        assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max(), key
    assert out['auto']['power_local_rows'] == out['gemm']['power_local_rows']



def test_transposed_result_buffer_grows_and_moves_without_changing_results(ma):
    """the transposed stage-1 result lives in physical pieces mapped into a reserved address range (common.h
    DevBuf::piece): it GROWS inside the range, and a result that outgrows the range moves to a new one (the old
    range is never handed back: on this runtime a re-used range reads stale data).  Sizes chosen to walk that path -
    16, 67 and 269 MB of result, then small again - each against the GEMMs on the same fields"""
    from metalens_amd import _lib
    c = _lib.default_context()
    c.set_precision('f64')
    try:
        for n, seed in ((512, 1), (1024, 2), (2048, 3), (768, 4), (512, 5)):
            p = WL / 2.2
            x = (np.arange(n) - (n - 1) / 2) * p
            u = lattice(n, x[1] - x[0], n, -(n // 2))
            F = fields(n, n, seed)
            c.set_method('fft-streamed')
            a = ma.farfield_direct(*F, x, x, WL, N_GLASS, u, u, ctx=c)
            assert c.plan_kernels() == ('fft', 'fft')
            c.set_method('gemm')
            b = ma.farfield_direct(*F, x, x, WL, N_GLASS, u, u, ctx=c)
            for key in ('Nx', 'Ny', 'Lx', 'Ly'):
                assert np.abs(a[key] - b[key]).max() <= 1e-12 * np.abs(b[key]).max(), (n, key)
    finally:
        c.set_method('auto')
