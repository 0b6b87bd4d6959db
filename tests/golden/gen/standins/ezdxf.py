"""Inert stand-in (module-level import in design_collimator.py only)."""
