"""Seeded Needleman-Wunsch pairs (kept in hipstr_amd/gen.py so that bench.py can use them too)."""
from hipstr_amd.gen import nw_pairs  # noqa: F401
