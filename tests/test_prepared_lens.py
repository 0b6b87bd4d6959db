"""ma.PreparedLens (metalens_amd/prepared.py): the host logic without a GPU - tokens decide
whether anything is hashed or uploaded again."""
import math

import pytest

import metalens_amd as ma
from metalens_amd import layout, packing, prepared, synthetic


class _Lib:
    def __init__(self):
        self.tables = 0
        self.layouts = 0

    def ml_upload_table(self, *a):
        self.tables += 1
        return 0

    def ml_upload_layout(self, *a):
        self.layouts += 1
        return 0


class _Ctx:
    def __init__(self):
        self.lib = _Lib()
        self.handle = None
        self.tables_token = None
        self.layout_token = None
        self.tables_fingerprint = None


def _lens(radius=15e-6):
    return synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet), layout.make_design,
                               radius=radius, numerical_aperture=0.3, wavelength=580e-9,
                               switch_angle=8 * math.pi / 180, num_gratings=6, num_entries=6)


def test_prepared_lens_hashes_once_and_follows_the_context(monkeypatch):
    a, b = _lens(), _lens(radius=16e-6)
    ctx = _Ctx()
    hashed = []
    real = packing._tables_fingerprint
    monkeypatch.setattr(packing, '_tables_fingerprint', lambda *args: hashed.append(1) or real(*args))
    pa = prepared.PreparedLens(a['lens_periphery_summary'], a['lens_center_summary'], a['hexgridset'], 580e-9, ctx=ctx)
    assert len(hashed) == 1 and ctx.lib.layouts == 1 and ctx.lib.tables > 0
    n_tables = ctx.lib.tables
    for _ in range(5):                      # repeated calls: two token compares, nothing else
        pa.make_resident(ctx, 580)
    assert len(hashed) == 1 and ctx.lib.layouts == 1 and ctx.lib.tables == n_tables
    pb = prepared.PreparedLens(b['lens_periphery_summary'], b['lens_center_summary'], b['hexgridset'], 580e-9, ctx=ctx)
    assert ctx.lib.layouts == 2 and pb.tokens != pa.tokens
    pa.make_resident(ctx, 580)              # the other lens took the context: this one goes up again
    assert ctx.lib.layouts == 3 and (ctx.tables_token, ctx.layout_token) == pa.tokens
    with pytest.raises(ValueError):
        pa.make_resident(ctx, 450)          # prepared for another wavelength
    with pytest.raises(ValueError):
        pa.make_resident(_Ctx(), 580)       # ... or on another context
