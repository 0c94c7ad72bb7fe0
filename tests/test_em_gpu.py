"""GPU: hipstr_em_train (EM stutter training, all loci in lock step on the device) against the compiled reference's golden vectors
and the oracle.  Tolerance: the E-step accumulation and the seven M-step reductions are bit-exact given their inputs, the exact
log-sum-exps use the device's exp/log (posteriors ~1e-13); stated |d| <= 1e-9 on parameters and 1e-9 relative on the
log-likelihood, identical iteration counts and train() results (observed: 1e-15 / 1e-14)."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
from em_cases import em_case
from test_em_oracle import FIXTURES, load

pytestmark = pytest.mark.gpu


def _same(got, want):
    return (np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.all(np.abs(got[1] - want[1]) <= 1e-9)
            and np.all(np.abs(got[3] - want[3]) <= 1e-9 * np.maximum(1, np.abs(want[3]))))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_golden_fixtures(hmm, path):
    kw, d = load(path)
    got = capi.run_em(hmm, "hipstr_", **kw)
    assert _same(got, (d["expect_trained"], d["expect_stutter"], d["expect_n_iter"], d["expect_final_ll"]))


def test_config3_shape_against_oracle(hmm, oracle):
    """BASELINE configs[2]: ~100 samples at low depth per locus; a few loci for the oracle, the same loci inside a larger batch."""
    kw = em_case(77, n_loci=40, samples=(90, 100), reads_per_sample=(3, 7))
    got = capi.run_em(hmm, "hipstr_", **kw)
    n = 4
    cut = {k: (np.asarray(v)[:n] if k in ("period", "n_samples", "haploid") else v) for k, v in kw.items()}
    cut["read_off"] = np.asarray(kw["read_off"])[:n + 1]
    for k in ("sample_label", "num_bps", "log_p1", "log_p2"):
        cut[k] = np.asarray(kw[k])[:cut["read_off"][-1]]
    want = capi.run_em(oracle, "oracle_", **cut)
    assert _same(tuple(x[:n] for x in got), want)
    assert got[0].all() and np.all(got[1][:, [0, 3]] <= 0.999) and np.all(got[1] > 0)


def test_many_loci_increasing_allele_counts(hmm, oracle):
    """Every locus of a batch whose allele counts grow from locus to locus, compared with the oracle.  The M-step scratch of a
    locus (row log-sum-exps) must not overlap its neighbours' whatever their allele counts (one workgroup per locus, all
    concurrent) — an overlap shows up as run-to-run differences in the allele-frequency priors, so the batch is also run twice."""
    counts = [2 + (l * 13) // 47 for l in range(48)]
    kw = em_case(99, n_loci=48, samples=(14, 16), reads_per_sample=(3, 5), allele_counts=counts)
    got = capi.run_em(hmm, "hipstr_", **kw)
    again = capi.run_em(hmm, "hipstr_", **kw)
    assert all(np.array_equal(a, b) for a, b in zip(got, again))
    want = capi.run_em(oracle, "oracle_", **kw)
    assert _same(got, want)


def test_empty_batch_and_errors(hmm):
    assert all(len(x) == 0 for x in capi.run_em(hmm, "hipstr_", [], [], [0], [], [], [], []))
    with pytest.raises(RuntimeError, match="ascending sample"):
        capi.run_em(hmm, "hipstr_", [4], [2], [0, 3], [1, 0, 1], [0, 4, 4], [0, 0, 0], [0, 0, 0])
