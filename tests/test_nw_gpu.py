"""GPU: hipstr_nw_align (systolic Needleman-Wunsch + pointer walk) against the compiled reference's golden vectors and the oracle.
Scores are exact float sums, so everything — score, gapped strings, CIGAR — is compared for equality."""
import os

import pytest

from hipstr_amd import capi
from nw_cases import nw_pairs
from test_nw_oracle import FIXTURES, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_golden_fixtures(hmm, path):
    pairs, pen, exp = load(path)
    assert capi.run_nw(hmm, "hipstr_", pairs, pen) == exp


@pytest.mark.parametrize("kw,pen", [(dict(n=80), False), (dict(n=80), True), (dict(n=60, read_len=(1, 256)), False),
                                    (dict(n=40, ref_len=(1500, 3000), read_len=(200, 256)), True),
                                    (dict(n=30, ref_len=(1400, 1800), read_len=(257, 1400)), True)])     # haplotype-sized: 5-24 rows per lane
def test_matches_oracle_on_seeded_pairs(hmm, oracle, kw, pen):
    pairs = nw_pairs(41, **kw)
    assert capi.run_nw(hmm, "hipstr_", pairs, pen) == capi.run_nw(oracle, "oracle_", pairs, pen)


def test_independent_of_the_workspace_chunking(hmm, monkeypatch):
    pairs = nw_pairs(42, n=120)
    whole = capi.run_nw(hmm, "hipstr_", pairs, False)
    monkeypatch.setenv("HIPSTR_NW_WS_MIB", "1")
    assert capi.run_nw(hmm, "hipstr_", pairs, False) == whole


def test_errors(hmm):
    with pytest.raises(RuntimeError, match="longer than 1536"):
        capi.run_nw(hmm, "hipstr_", [("ACGT" * 100, "A" * 1600)], False)
    with pytest.raises(RuntimeError, match="non-empty"):
        capi.run_nw(hmm, "hipstr_", [("ACGT", "")], False)
    assert capi.run_nw(hmm, "hipstr_", [], False) == []
