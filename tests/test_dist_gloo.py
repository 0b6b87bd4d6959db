"""The N>1 decomposition on CPU: world_size-2 processes (gloo) each take their block of
aperture rows (metalens_amd.dist.row_block), compute the near field and the partial radiation
vectors of those rows with the CPU oracle, all-reduce, and must reproduce the unsharded
result.  This is exactly the data flow of the GPU path (near field -> partial N,L -> one
all-reduce(sum) -> projection) with gloo standing in for RCCL and the oracle for the kernels."""
import os
import socket

import numpy as np
import pytest

import golden_io
from metalens_amd import dist


def test_mirrored_blocks_cover_everything():
    for n in (16, 400, 2048, 5792):
        for world in (1, 2, 3, 8):
            seen = np.concatenate([dist.mirrored_rows(n, *dist.mirrored_block(n, world, r))
                                   for r in range(world)])
            assert np.array_equal(np.sort(seen), np.arange(n))
            for r in range(world):
                rows = dist.mirrored_rows(n, *dist.mirrored_block(n, world, r))
                assert np.array_equal(rows + rows[::-1], np.full(rows.size, n - 1))


def test_weighted_mirrored_blocks():
    n, world = 2048, 8
    x = (np.arange(n // 2) - (n - 1) / 2) * 1.0
    w = 3.2 + 0.8 * np.minimum(2 * np.sqrt(np.maximum(700.0 ** 2 - x ** 2, 0)) / n, 1.0)
    blocks = [dist.mirrored_block(n, world, r, weights=w) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n // 2
    assert all(a[1] == b[0] for a, b in zip(blocks[:-1], blocks[1:]))
    cost = [w[a:b].sum() for a, b in blocks]
    assert max(cost) / (sum(cost) / world) < 1.05        # balanced to the 8-row granularity
    sizes = [b - a for a, b in blocks]
    assert sizes[0] > sizes[-1]                          # rim ranks get more (cheaper) rows


def test_row_block_partitions_cover_everything():
    for n in (1, 15, 16, 400, 2048, 2897, 8192):
        for world in (1, 2, 3, 4, 8):
            blocks = [dist.row_block(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a0, a1), (b0, b1) in zip(blocks[:-1], blocks[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in blocks]
            if n >= 64 * world:
                assert max(sizes) - min(sizes) <= 32
                assert all(a % 16 == 0 for a, _ in blocks)


def test_interleaved_rows_cover_everything():
    for n, world, block in ((16, 2, 1), (2048, 8, 1), (8192, 8, 4), (4096, 4, 4), (512, 2, 2)):
        shards = [dist.interleaved_rows(n, world, r, block) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(n))
        for r, rows in enumerate(shards):
            assert rows.size == n // world
            assert np.array_equal(rows[:block], block * r + np.arange(block))
            assert np.array_equal(np.diff(rows.reshape(-1, block), axis=0),
                                  np.full((rows.size // block - 1, block), block * world))


@pytest.mark.parametrize('N,G,s,M,j0', [(1024, 2, 8, 64, -32), (8192, 8, 8, 512, -256), (4096, 1, 2, 512, -256),
                                        ((1024, 960), 4, 8, 64, -32), ((4096, 4000), 8, 4, 512, -256), (3072, 1, 3, 100, 1500), (1024, 4, 1, 64, -32), (2048, 8, 1, 100, -37), (4096, 4, 4, 512, -256),
                                        (1024, 2, 2, 1024, -512), (2048, 2, 4, 96, 900)])
def test_interleaved_shard_column_pass_is_a_short_dft(N, G, s, M, j0):
    """What csrc/farfield.hip's interleaved column pass computes, in NumPy: rank r's partial sum
    over its rows n = s (G m + r) + i on the bins k_j = j + j0 of the FULL N-point lattice is, per
    i, an (N / sG)-point DFT over m read at bin k_j mod (N / sG), times exp(2 pi i (c - s r - i) k_j / N)
    (zfft_interleave_tables_kernel).  The ranks' partial sums add up to the whole aperture sum
    sum_n x[n] exp(-2 pi i (n - c) k_j / N)  (nearfield_farfield.py:111-120)."""
    # (N, rows): a lattice longer than the aperture - the rows beyond `rows` are zeros nobody holds
    N, n_rows = N if isinstance(N, tuple) else (N, N)
    rng = np.random.default_rng(N + G + s)
    x = rng.standard_normal(n_rows) + 1j * rng.standard_normal(n_rows)
    c = n_rows - n_rows // 2
    k = np.arange(M) + j0
    whole = np.array([np.sum(x * np.exp(-2j * np.pi * ((np.arange(n_rows) - c) * kj % N) / N)) for kj in k])
    Nsub = N // (s * G)
    total = np.zeros(M, dtype=complex)
    for r in range(G):
        rows = dist.interleaved_rows(n_rows, G, r, s)
        local = x[rows]                                   # resident order
        for i in range(s):
            sub = np.fft.fft(local[i::s], n=Nsub)         # Nsub points (zero-padded), sG apart in the aperture
            assert sub.size == Nsub
            pj = np.exp(2j * np.pi * (((c - s * r - i) * k) % N) / N)
            total += sub[k % Nsub] * pj
    assert np.abs(total - whole).max() <= 1e-12 * np.abs(whole).max()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, sharding='mirrored'):
    import torch
    import torch.distributed as td
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import farfield_oracle, nearfield_oracle
    case = np.load(golden_io.golden_path('nearfield_B_straddle_offaxis_y.npz'))
    lens = golden_io.load_lens(golden_io.golden_path(str(case['lens'])))
    x, y = case['x_pts'], case['y_pts']
    wl = float(case['wavelength'])
    # the decompositions HotPath uses for N > 1: blocks of rows dealt round robin (direction grids
    # on the FFT lattice) or mirrored row pairs per rank
    if sharding == 'interleaved':
        rows = dist.interleaved_rows(len(x), world, rank, 2)
        runs = rows.reshape(-1, 2)
    else:
        q0, q1 = dist.mirrored_block(len(x), world, rank, align=2)
        rows = dist.mirrored_rows(len(x), q0, q1)
        runs = (rows[:q1 - q0], rows[q1 - q0:])
    # (the near-field oracle checks that x_pts is uniform, so feed it the runs separately)
    parts = [nearfield_oracle.build_nearfield(
        float(case['source_x']), float(case['source_y']), float(case['source_z']),
        str(case['source_pol']), wl, lens[0], lens[1], lens[2], x_pts=x[run], y_pts=y,
        c0=float(case['c0']), Z0=float(case['Z0'])) for run in runs]
    Ex, Ey, Hx, Hy = (np.vstack([p[k] for p in parts]) for k in range(4))
    power = sum(p[6] for p in parts)
    n_glass = parts[0][7]
    ux = np.linspace(-0.3, 0.5, 9)
    uy = np.linspace(-0.2, 0.2, 7)
    part = farfield_oracle.radiation_vectors(Ex, Ey, Hx, Hy, x, y, wl, n_glass, ux, uy,
                                             row_range=rows)
    # RCCL has no complex type: reduce as float64 pairs, exactly like ml_farfield_allreduce
    buf = torch.from_numpy(np.stack(part).view(np.float64).copy())
    td.all_reduce(buf, op=td.ReduceOp.SUM)
    p = torch.tensor([power], dtype=torch.float64)
    td.all_reduce(p, op=td.ReduceOp.SUM)
    if rank == 0:
        np.savez(os.path.join(out_dir, 'reduced.npz'), vectors=buf.numpy().view(np.complex128),
                 power=p.numpy(), ux=ux, uy=uy)
    td.destroy_process_group()


@pytest.mark.parametrize('sharding', ['mirrored', 'interleaved'])
def test_two_rank_sharded_sum_equals_whole(tmp_path, sharding):
    torch = pytest.importorskip('torch')
    import torch.multiprocessing as mp
    from oracle import farfield_oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), sharding), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), 'reduced.npz'))
    case = np.load(golden_io.golden_path('nearfield_B_straddle_offaxis_y.npz'))
    whole = farfield_oracle.radiation_vectors(case['Ex'], case['Ey'], case['Hx'], case['Hy'],
                                              case['x_pts'], case['y_pts'],
                                              float(case['wavelength']), float(case['n_glass']),
                                              z['ux'], z['uy'])
    for got, want in zip(z['vectors'], whole):
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
    assert abs(z['power'][0] - case['power']) <= 1e-13 * abs(case['power'])


def test_rank_blocked_amplitudes_round_trip():
    rng = np.random.default_rng(5)
    for world, mx, my in ((2, 8, 5), (4, 512, 512), (8, 64, 1)):
        a = rng.standard_normal((mx, my)) + 1j * rng.standard_normal((mx, my))
        b = rng.standard_normal((mx, my)) + 1j * rng.standard_normal((mx, my))
        buf = dist.block_amplitudes(a, b, world)
        assert buf.size == 2 * 2 * mx * my and buf.size % world == 0
        # chunk r = rows [r R, (r + 1) R) of a_theta, then the same rows of a_phi
        rows, chunk = mx // world, buf.size // world
        for r in range(world):
            c = buf[r * chunk:(r + 1) * chunk].view(np.complex128).reshape(2, rows, my)
            assert np.array_equal(c[0], a[r * rows:(r + 1) * rows]) and np.array_equal(c[1], b[r * rows:(r + 1) * rows])
        ga, gb = dist.unblock_amplitudes(buf, world, (mx, my))
        assert np.array_equal(ga, a) and np.array_equal(gb, b)


def _scatter_worker(rank, world, port, out_dir):
    """the step's collective as the GPU path runs it (csrc/farfield.hip ml_farfield_project_reduce /
    ml_farfield_gather), with gloo standing in for RCCL: every rank projects ITS partial amplitudes into
    the rank-blocked buffer, a reduce-scatter leaves rank r with the sum of block r (gloo has no
    reduce-scatter: one reduce per root), each rank takes the power of its block, an all-gather of
    blocks and power rows completes every rank's picture"""
    import torch
    import torch.distributed as td
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    mx, my = 16, 6
    rng = np.random.default_rng(100 + rank)
    a_theta = rng.standard_normal((mx, my)) + 1j * rng.standard_normal((mx, my))   # this rank's partial sums
    a_phi = rng.standard_normal((mx, my)) + 1j * rng.standard_normal((mx, my))
    buf = torch.from_numpy(dist.block_amplitudes(a_theta, a_phi, world))
    chunk = buf.numel() // world
    for root in range(world):     # reduce-scatter
        part = buf[root * chunk:(root + 1) * chunk].clone()
        td.reduce(part, dst=root, op=td.ReduceOp.SUM)
        if root == rank:
            mine = part
    rows = mx // world
    blk = mine.numpy().view(np.complex128).reshape(2, rows, my)
    power = np.abs(blk[0]) ** 2 + np.abs(blk[1]) ** 2          # (stands for nearfield_farfield.py:184-189 on the block)
    gathered = [torch.empty(chunk, dtype=torch.float64) for _ in range(world)]
    td.all_gather(gathered, mine)
    pw = [torch.empty(rows * my, dtype=torch.float64) for _ in range(world)]
    td.all_gather(pw, torch.from_numpy(power.ravel().copy()))
    ga, gb = dist.unblock_amplitudes(torch.cat(gathered).numpy(), world, (mx, my))
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), a_theta=a_theta, a_phi=a_phi, sum_theta=ga, sum_phi=gb,
             power=torch.cat(pw).numpy().reshape(mx, my))
    td.destroy_process_group()


def test_two_rank_reduce_scatter_of_blocked_amplitudes(tmp_path):
    pytest.importorskip('torch')
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_scatter_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(2)]
    want_t, want_p = z[0]['a_theta'] + z[1]['a_theta'], z[0]['a_phi'] + z[1]['a_phi']
    for r in range(2):
        assert np.array_equal(z[r]['sum_theta'], want_t) and np.array_equal(z[r]['sum_phi'], want_p)
        assert np.array_equal(z[r]['power'], np.abs(want_t) ** 2 + np.abs(want_p) ** 2)
