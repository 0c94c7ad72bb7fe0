"""CPU: the EM restatement (oracle_em_train = EMStutterGenotyper::train and everything it calls) against golden vectors of the
compiled reference: same libm, same operation order -> identical parameters, iteration counts and log-likelihoods."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
from em_cases import em_case

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "em_*.npz")))
KEYS = ("period", "n_samples", "read_off", "sample_label", "num_bps", "log_p1", "log_p2", "haploid")


def load(path):
    d = np.load(path)
    kw = {k: d[k] for k in KEYS}
    if "max_iter" in d.files:
        kw["max_iter"] = int(d["max_iter"])
    return kw, d


def test_fixtures_present():
    assert len(FIXTURES) >= 5


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_oracle_matches_golden(oracle, path):
    kw, d = load(path)
    tr, st, it, ll = capi.run_em(oracle, "oracle_", **kw)
    assert np.array_equal(tr, d["expect_trained"]) and np.array_equal(it, d["expect_n_iter"])
    assert np.array_equal(st, d["expect_stutter"]) and np.array_equal(ll, d["expect_final_ll"])


def test_golden_has_a_failed_training():
    """max_iter = 3 must leave some locus untrained (train() returning false) so that path is pinned too."""
    _, d = load(os.path.join(HERE, "golden", "em_few_iter.npz"))
    assert not d["expect_trained"].all() and d["expect_n_iter"].max() == 3


@pytest.mark.skipif(not os.path.exists(capi.REF_LIB), reason="compiled reference (oracle/_ref) not built")
def test_oracle_matches_compiled_reference_on_fresh_cases(oracle):
    ref = capi.load_ref()
    for seed in range(20, 26):
        kw = em_case(seed, n_loci=4)
        a = capi.run_em(ref, "ref_", **kw); b = capi.run_em(oracle, "oracle_", **kw)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
