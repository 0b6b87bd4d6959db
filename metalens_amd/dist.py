"""Multi-GPU plumbing: one process per GPU, RCCL over xGMI through the C ABI.

What is distributed: the aperture ROWS (x index).  Near-field
synthesis is independent per sample and the far-field sum is linear in the aperture
(the reason the reference's own strip chunking is exact, nearfield.py:482-516), so
each rank synthesises and transforms its own rows with no communication, and the
partial radiation vectors ``Nx,Ny,Lx,Ly [mx][my]`` are summed with ONE all-reduce
before the (non-linear) projection.  Rows rather than radial annuli: every rank gets
the same GEMM shape, and contiguous rows are one contiguous block of the C-ordered
field arrays.

No PyTorch here: ranks find each other through the environment variables that
``python -m torch.distributed.run`` exports (RANK, LOCAL_RANK, WORLD_SIZE,
MASTER_PORT) and the 128-byte RCCL unique id travels through a file in /tmp (all
ranks are on one node).
"""
import os
import time

import numpy as np

from . import _lib


def force_rccl():
    """ML_FORCE_RCCL=1: build a real one-rank RCCL communicator and run the all-reduce even
    with a single GPU (test hook for the multi-GPU code path)"""
    return os.environ.get('ML_FORCE_RCCL', '0') not in ('', '0')


def env_rank():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def row_block(n_rows, world, rank, align=16):
    """rows [r0, r1) of rank ``rank``: contiguous, as equal as possible, block edges
    aligned to ``align`` rows where that is possible (the GEMM tiles are 16 high)"""
    if world == 1:
        return 0, n_rows
    edges = [min(n_rows, int(round(n_rows * k / world / align)) * align) for k in range(world + 1)]
    edges[0], edges[-1] = 0, n_rows
    for k in range(1, world + 1):
        edges[k] = max(edges[k], edges[k - 1])
    return edges[rank], edges[rank + 1]


def mirrored_block(n_rows, world, rank, align=8, weights=None):
    """Mirror-symmetric shard of an aperture with an EVEN number of rows: rank ``rank`` owns
    the row pairs q in [q0, q1), i.e. rows [q0, q1) and [n_rows-q1, n_rows-q0).  Every rank
    then holds both members of each +/-x' pair, which lets BOTH far-field stages run folded
    (csrc/zfold.hip).  Returns (q0, q1).

    ``weights[q]`` (optional, length n_rows/2) is the relative cost of row pair q; the pairs
    are then split so that every rank gets about the same total cost instead of the same
    count (rows through the lens centre cost more near-field time than rows at the rim)."""
    assert n_rows % 2 == 0
    half = n_rows // 2
    if weights is None or world == 1:
        return row_block(half, world, rank, align=align)
    import numpy as np
    w = np.asarray(weights, dtype=float)
    assert w.shape == (half,) and np.all(w > 0)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    edges = [0]
    for k in range(1, world):
        q = int(np.searchsorted(cum, cum[-1] * k / world))
        q = int(round(q / align)) * align
        edges.append(min(max(q, edges[-1]), half))
    edges.append(half)
    return edges[rank], edges[rank + 1]


def mirrored_rows(n_rows, q0, q1):
    """row indices of the shard (q0, q1), in resident order"""
    import numpy as np
    return np.concatenate((np.arange(q0, q1), np.arange(n_rows - q1, n_rows - q0)))


def interleaved_rows(n_rows, world, rank, block):
    """row indices of rank ``rank`` when blocks of ``block`` rows are dealt round robin over
    ``world`` ranks (rows ``block * (world * m + rank) + i``), in resident order; needs
    ``n_rows`` to be a multiple of ``block * world``"""
    import numpy as np
    assert n_rows % (block * world) == 0
    m = np.arange(n_rows // (block * world))
    return (block * (world * m[:, None] + rank) + np.arange(block)[None, :]).ravel()


def block_amplitudes(a_theta, a_phi, world):
    """the two projected amplitudes [mx][my] in the RANK-BLOCKED order the GPU path reduces them in
    (csrc/farfield.hip ProjArgs::blk_rows): block b = direction rows [b R, (b + 1) R), R = mx / world,
    of BOTH planes, contiguous - what a reduce-scatter hands rank b.  Returns a float64 vector (RCCL has
    no complex type) of world equal chunks."""
    a_theta, a_phi = np.asarray(a_theta, dtype=np.complex128), np.asarray(a_phi, dtype=np.complex128)
    mx = a_theta.shape[0]
    assert a_theta.shape == a_phi.shape and mx % world == 0
    rows = mx // world
    blocks = np.stack([a_theta.reshape(world, rows, -1), a_phi.reshape(world, rows, -1)], axis=1)   # [b][plane][R][my]
    return np.ascontiguousarray(blocks).view(np.float64).ravel()


def unblock_amplitudes(buf, world, shape):
    """inverse of block_amplitudes: (a_theta, a_phi) of ``shape`` from the rank-blocked vector"""
    mx = shape[0]
    rows = mx // world
    blocks = np.asarray(buf, dtype=np.float64).view(np.complex128).reshape(world, 2, rows, -1)
    return blocks[:, 0].reshape(shape), blocks[:, 1].reshape(shape)


def _id_path():
    key = '%s_%s_%s' % (os.environ.get('MASTER_PORT', '0'),
                        os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.getppid())
    return os.path.join(os.environ.get('TMPDIR', '/tmp'), 'metalens_rccl_id_%s.bin' % key)


def exchange_unique_id(rank, world, timeout=120.0, path=None):
    """rank 0 creates the RCCL unique id and publishes it atomically; the others poll"""
    path = path or _id_path()
    lib = _lib.load()
    if rank == 0:
        buf = (_lib.c_uint8 * 128)()
        _lib.check(lib.ml_comm_unique_id(buf))
        tmp = path + '.tmp%d' % os.getpid()
        with open(tmp, 'wb') as f:
            f.write(bytes(buf))
        os.replace(tmp, path)
        return bytes(buf)
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                data = f.read()
            if len(data) == 128:
                return data
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise _lib.MetalensHipError('timed out waiting for the RCCL unique id at ' + path)
        time.sleep(0.05)


def init_comm(ctx, rank, world):
    if world <= 1 and not force_rccl():
        return
    uid = exchange_unique_id(rank, world)
    buf = (_lib.c_uint8 * 128).from_buffer_copy(uid)
    _lib.check(ctx.lib.ml_comm_init(ctx.handle, buf, world, rank))
    barrier(ctx)
    if rank == 0:
        try:
            os.remove(_id_path())
        except OSError:
            pass


def barrier(ctx):
    _lib.check(ctx.lib.ml_comm_barrier(ctx.handle))


def allreduce_host(ctx, values, op='sum'):
    a = np.ascontiguousarray(values, dtype=np.float64).copy()
    _lib.check(ctx.lib.ml_comm_allreduce_host(ctx.handle, _lib.dptr(a), a.size,
                                              {'sum': 0, 'max': 1}[op]))
    return a
