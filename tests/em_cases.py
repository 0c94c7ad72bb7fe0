"""Seeded inputs for the de novo stutter EM (kept in hipstr_amd/gen.py so that bench.py can use them too)."""
from hipstr_amd.gen import em_case  # noqa: F401
