"""CPU: the Needleman-Wunsch restatement (oracle_nw_align = NeedlemanWunsch::Align) against golden vectors of the compiled
reference: score, both gapped strings and the CIGAR must be identical."""
import glob
import json
import os

import numpy as np
import pytest

from hipstr_amd import capi
from nw_cases import nw_pairs

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "nw_*.npz")))


def load(path):
    d = np.load(path)
    pairs = [tuple(x) for x in json.loads(bytes(d["pairs"].tobytes()).decode())]
    exp = [tuple(x) for x in json.loads(bytes(d["expect"].tobytes()).decode())]
    return pairs, bool(d["end_penalty"][0]), exp


def test_fixtures_present():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_oracle_matches_golden(oracle, path):
    pairs, pen, exp = load(path)
    assert capi.run_nw(oracle, "oracle_", pairs, pen) == exp


def test_alignment_strings_are_consistent(oracle):
    """Property: stripping the gaps gives the inputs back; the CIGAR spends the read exactly."""
    import re
    pairs = nw_pairs(11, n=30)
    for (ref, read), (score, ok, ra, qa, cig) in zip(pairs, capi.run_nw(oracle, "oracle_", pairs, False)):
        assert ok and ra.replace("-", "") == ref and qa.replace("-", "") == read and len(ra) == len(qa)
        assert sum(int(n) for n, op in re.findall(r"(\d+)([=XI])", cig)) == len(read)


@pytest.mark.skipif(not os.path.exists(capi.REF_LIB), reason="compiled reference (oracle/_ref) not built")
def test_oracle_matches_compiled_reference_on_fresh_cases(oracle):
    ref = capi.load_ref()
    for seed in range(30, 34):
        for pen in (False, True):
            pairs = nw_pairs(seed, n=25)
            assert capi.run_nw(oracle, "oracle_", pairs, pen) == capi.run_nw(ref, "ref_", pairs, pen)
