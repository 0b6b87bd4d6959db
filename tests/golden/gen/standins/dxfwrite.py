"""Inert stand-in: the reference's CAD export module-level import only
(design_collimator.py:20-24); never executed on the near-field path."""


class DXFEngine:  # pragma: no cover
    pass
