"""A lens prepared once for many drop-in calls.

``build_nearfield`` (nearfield.py) takes the reference's own objects (nearfield.py:76-80 of the
reference: ``lens_periphery_summary``, ``lens_center_summary``, ``hexgridset``) and has to decide on
EVERY call whether the tables and the layout on the GPU are still these - a content hash of the
caller's arrays, 1.8 ms for the 1 mm lens and 30 MB of cells for a millimetre-scale one.  A caller
who sweeps sources over one lens knows they are:

    lens = ma.PreparedLens(lens_periphery_summary, lens_center_summary, hexgridset, wavelength)
    for src in sources:
        out = ma.build_nearfield(*src, wavelength, lens, None, None, x_pts=x, y_pts=y)

The hashes are taken (and the uploads made) once, in the constructor; later calls compare two tokens
with the context's and go straight to the synthesis.  If ANOTHER lens has used the context in
between, the tokens differ and this lens is uploaded again (hashed again, once).  The contract the
caller accepts: the arrays are not edited in place while the handle is in use - call ``refresh()``
after editing."""
import numpy as np

from . import _lib, constants, packing


class PreparedLens:
    def __init__(self, lens_periphery_summary, lens_center_summary, hexgridset, wavelength, ctx=None, units=None):
        self.lens_periphery_summary = lens_periphery_summary
        self.lens_center_summary = lens_center_summary
        self.hexgridset = hexgridset
        self.wavelength = wavelength
        # (``units``: the caller's unit system, as for build_nearfield - what a nanometre is decides the table key)
        self.wavelength_in_nm = int(round(wavelength / constants.as_units(units).nm))
        self.ctx = ctx or _lib.default_context()
        self.tokens = None
        self.refresh()

    def refresh(self):
        """hash the caller's arrays again and upload whatever changed"""
        S = self.lens_periphery_summary
        packing.upload_tables(self.ctx, S['gratingcollection_list'], self.hexgridset, self.wavelength_in_nm)
        packing.upload_layout(self.ctx, S, self.lens_center_summary)
        self.tokens = (self.ctx.tables_token, self.ctx.layout_token)

    def make_resident(self, ctx, wavelength_in_nm):
        """no-op when ``ctx`` still holds this lens; otherwise one full (hashed) upload"""
        if wavelength_in_nm != self.wavelength_in_nm:
            raise ValueError('this lens was prepared for %d nm, not %d nm' % (self.wavelength_in_nm, wavelength_in_nm))
        if ctx is not self.ctx:
            raise ValueError('a PreparedLens belongs to the context it was prepared on')
        if (ctx.tables_token, ctx.layout_token) != self.tokens or self.tokens[0] is None:
            self.refresh()
