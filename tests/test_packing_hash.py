"""Host logic that decides whether resident tables / layouts can be re-used: the content hashes
taken from the caller's own arrays before anything is packed (metalens_amd/packing.py), and the
grouping of a source list into polarisation batches (metalens_amd/sweep.py).  No GPU."""
import math

import numpy as np

import metalens_amd as ma
from metalens_amd import layout, packing, synthetic


def _lens():
    return synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet), layout.make_design,
                               radius=15e-6, numerical_aperture=0.3, wavelength=580e-9,
                               switch_angle=8 * math.pi / 180, num_gratings=6, num_entries=6)


def test_tables_fingerprint_follows_content_not_identity():
    lens = _lens()
    objs = list(lens['lens_periphery_summary']['gratingcollection_list']) + [lens['hexgridset']]
    a = packing._tables_fingerprint(objs, 580)
    assert a == packing._tables_fingerprint(objs, 580)                 # deterministic
    assert a != packing._tables_fingerprint(objs[:-1] + [None], 580)   # centre set missing
    # one value of one table edited IN PLACE: same objects, another fingerprint
    key = sorted(objs[0].interpolators, key=repr)[3]
    vals = objs[0].interpolators[key].values
    old = vals.flat[17]
    vals.flat[17] = old * (1 + 1e-15) + 1e-30
    b = packing._tables_fingerprint(objs, 580)
    vals.flat[17] = old
    assert b != a and packing._tables_fingerprint(objs, 580) == a
    # another wavelength has no tables at all
    assert packing._tables_fingerprint(objs, 450) != a


class _FakeCtx:
    """records what upload_layout would send, without a GPU"""
    def __init__(self):
        self.layout_token = None
        self.uploads = 0

        class Lib:
            def ml_upload_layout(_self, *args):
                self.uploads += 1
                return 0
        self.lib = Lib()
        self.handle = None


def test_layout_token_sees_reordered_cells_and_skips_identical_calls():
    lens = _lens()
    S, cells = lens['lens_periphery_summary'], np.array(lens['lens_center_summary'], dtype=float)
    ctx = _FakeCtx()
    packing.upload_layout(ctx, S, cells)
    first = ctx.layout_token
    packing.upload_layout(ctx, S, cells.copy())            # equal content, other object: no upload
    assert ctx.uploads == 1 and ctx.layout_token == first
    perm = np.random.default_rng(0).permutation(len(cells))
    packing.upload_layout(ctx, S, cells[perm])             # same cells, another order: a new layout
    assert ctx.uploads == 2 and ctx.layout_token != first
    swapped = cells.copy()
    swapped[[0, 1], 2] = swapped[[1, 0], 2]                # two cell types exchanged (sums unchanged)
    if swapped[0, 2] != cells[0, 2]:
        packing.upload_layout(ctx, S, swapped)
        assert ctx.uploads == 3
    S2 = dict(S)
    S2['r_center_list'] = np.array(S['r_center_list'], dtype=float) * (1 + 1e-12)
    packing.upload_layout(ctx, S2, cells)                  # r_center feeds xp and the lateral period
    assert ctx.layout_token not in (first,)


def test_sweep_groups_consecutive_polarisations_of_one_position():
    from metalens_amd.sweep import MAX_BATCH, SourceSweep
    sw = SourceSweep.__new__(SourceSweep)                  # the grouping needs no GPU state
    f = -1e-4
    srcs = [(0, 0, f, 'x'), (0, 0, f, 'y'), (0, 0, f, 'z'), (0, 0, f, 'x'),      # 3 + 1 (batch is full)
            (1e-6, 0, f, 'x'), (0, 0, f, 'y'), (0, 0, f, 'z'),                      # position changes
            (0, 0, -float('inf'), 'x'), (0, 0, -float('inf'), 'y')]                # plane waves
    groups = sw._group(srcs)
    # (the two single sources at different positions travel as one position batch)
    assert [len(g['members']) for g in groups] == [3, 2, 2, 2]
    assert groups[1].get('mixed') and groups[1]['positions'] == [(0, 0, f), (1e-6, 0, f)]
    assert not groups[0].get('mixed') and not groups[2].get('mixed')
    assert all(len(g['members']) <= MAX_BATCH for g in groups)
    assert [k for g in groups for k, _ in g['members']] == list(range(len(srcs)))
    import pytest
    with pytest.raises(AssertionError):
        sw._group([(0, 0, -float('inf'), 'z')])            # no z-polarised plane wave (nearfield.py:224)
    with pytest.raises(AssertionError):
        sw._group([(0, 0, +1.0, 'x')])                     # the source sits below the lens


def test_large_arrays_are_hashed_in_parallel_chunks(monkeypatch):
    """the chunked hash (used where the host has the cores) is deterministic, sees every byte, and
    is only taken for arrays beyond the threshold"""
    monkeypatch.setattr(packing, '_BIG', 1 << 12)
    monkeypatch.setattr(packing, '_cores', lambda: 64)
    rng = np.random.default_rng(1)
    a = rng.standard_normal((4097, 3))

    def digest(arr):
        h = packing._hasher()
        packing._feed(h, arr)
        return h.digest()

    d = digest(a)
    assert d == digest(a.copy())
    for at in (0, 5000, a.size - 1):
        b = a.copy()
        b.flat[at] = np.nextafter(b.flat[at], np.inf)
        assert digest(b) != d
    small = rng.standard_normal(16)
    monkeypatch.setattr(packing, '_cores', lambda: 1)
    assert digest(small) == digest(small.copy()) and digest(a) == d   # few cores: same chunks hashed in line, same digest


def test_failed_table_upload_leaves_nothing_marked_resident():
    """an upload that fails part-way must not leave a fingerprint / token that a later call would
    take for 'these tables are on the GPU' """
    lens = _lens()
    gcs = lens['lens_periphery_summary']['gratingcollection_list']

    class Lib:
        def __init__(self):
            self.calls = 0
            self.fail_at = 2

        def ml_upload_table(self, *args):
            self.calls += 1
            return 3 if self.calls == self.fail_at else 0

        def ml_last_error(self):
            return b'simulated failure'

    class Ctx:
        handle = None
        tables_token = ('tables', 'something older')
        tables_fingerprint = b'older'

    ctx = Ctx()
    ctx.lib = Lib()
    import pytest
    with pytest.raises(Exception):
        packing.upload_tables(ctx, gcs, lens['hexgridset'], 580)
    assert ctx.tables_token is None and ctx.tables_fingerprint is None
    ctx.lib.fail_at = -1
    packing.upload_tables(ctx, gcs, lens['hexgridset'], 580)
    assert ctx.tables_token is not None and ctx.tables_fingerprint is not None
    n = ctx.lib.calls
    packing.upload_tables(ctx, gcs, lens['hexgridset'], 580)      # resident: no further upload
    assert ctx.lib.calls == n


def test_large_array_digest_does_not_depend_on_the_core_count(monkeypatch):
    """the chunk layout of a big array follows its byte count alone, so the token of a
    given content is the same whether the chunks are hashed on a pool or in line"""
    a = np.random.default_rng(3).standard_normal((packing._BIG // 8) + 1001)

    def digest():
        h = packing._hasher()
        packing._feed(h, a)
        return h.digest()

    monkeypatch.setattr(packing, '_cores', lambda: 1)
    serial = digest()
    monkeypatch.setattr(packing, '_cores', lambda: 64)
    pooled = digest()
    assert serial == pooled
    assert packing._POOL is not None
    packing._drop_pool()                      # what a forked child does
    assert packing._POOL is None and digest() == serial
    b = a.copy()
    b[-1] += 1.0
    h = packing._hasher()
    packing._feed(h, b)
    assert h.digest() != serial
