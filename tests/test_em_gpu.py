"""GPU: hipstr_em_train (EM stutter training; round 5: the iteration loop, the parameter update and the convergence tests run on the
device) against the compiled reference's golden vectors and the oracle.  Contract (util.assert_arrays_exact): train() results, iteration
counts, the six parameters and the final log-likelihood equal the reference BIT FOR BIT — every exp / log of the EM is the correctly
rounded function of cr_math.h, in the reference's operation order.  Where the host's libm is not correctly rounded on an argument of
the case, the second level has to explain the difference completely: device == the oracle run with the same correctly rounded functions,
bit for bit, and that within 1e-9 of the host-libm reference with identical iteration counts.  HIPSTR_EM_HOST_LOOP=1 (the round-4 loop
on the host with its libm) is kept and compared as well."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
from em_cases import em_case
from test_em_oracle import FIXTURES, load
import util

pytestmark = pytest.mark.gpu


def _same(got, want):
    return (np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.all(np.abs(got[1] - want[1]) <= 1e-9)
            and np.all(np.abs(got[3] - want[3]) <= 1e-9 * np.maximum(1, np.abs(want[3]))))


def _exact(got, want, oracle, kw, what):
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_em(oracle, "oracle_", **kw)
    return util.assert_arrays_exact(got, want, cr, what)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_golden_fixtures(hmm, oracle, path):
    kw, d = load(path)
    got = capi.run_em(hmm, "hipstr_", **kw)
    _exact(got, (d["expect_trained"], d["expect_stutter"], d["expect_n_iter"], d["expect_final_ll"]), oracle, kw, os.path.basename(path))


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
    _exact(tuple(x[:n] for x in got), want, oracle, cut, "configs[2] shape")
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
    _exact(got, want, oracle, kw, "48 loci, growing allele counts")


def test_more_allele_sizes_than_a_wavefront_has_lanes(hmm, oracle):
    """recalc_log_gt_priors (em_stutter_genotyper.cpp:22-57) on the device keeps one chain per allele size on the lanes of a wavefront, 64 at a
    time, with the chunk's values and exponentials in LDS tiles indexed by the alleles of the sweep — loci with 70 to 130 distinct sizes
    (highly polymorphic loci of a large cohort) take two sweeps and rows longer than a tile's usual stride; next to a small locus."""
    kw = em_case(321, n_loci=4, samples=(130, 150), reads_per_sample=(3, 4), haploid_rate=0.0, allele_counts=[70, 4, 100, 130])
    n_sizes = [len(set(kw["num_bps"][kw["read_off"][l]:kw["read_off"][l + 1]])) for l in range(4)]
    assert n_sizes[0] > 64 and n_sizes[2] > 64 and n_sizes[3] > 100, n_sizes
    got = capi.run_em(hmm, "hipstr_", **kw)
    want = capi.run_em(oracle, "oracle_", **kw)
    _exact(got, want, oracle, kw, "loci with more than 64 allele sizes")


def test_loci_that_converge_at_very_different_rounds(hmm, oracle, monkeypatch):
    """The device-resident loop compacts the list of loci still training after every round and sizes its launches by a count that lags a
    round behind: loci that stop after 2 rounds next to loci that need dozens, a locus that runs out of iterations (train() == false), and
    max_iter = 0 (no round at all) — each against the oracle, and the device loop against the round-4 host loop (HIPSTR_EM_HOST_LOOP=1:
    host libm, every locus launched in every round) within 1e-9 and with identical iteration counts."""
    kw = em_case(123, n_loci=60, samples=(4, 120), reads_per_sample=(2, 9))
    got = capi.run_em(hmm, "hipstr_", **kw)
    want = capi.run_em(oracle, "oracle_", **kw)
    _exact(got, want, oracle, kw, "60 loci, 4-120 samples")
    assert got[2].max() >= 3 * max(1, got[2].min())                 # the rounds really differ
    for mi in (0, 1, 3):
        kw2 = dict(kw); kw2["max_iter"] = mi
        g2 = capi.run_em(hmm, "hipstr_", **kw2); w2 = capi.run_em(oracle, "oracle_", **kw2)
        _exact(g2, w2, oracle, kw2, "max_iter = %d" % mi)
        assert np.all(g2[2] <= mi) and (mi > 2 or not g2[0].all())
    monkeypatch.setenv("HIPSTR_EM_HOST_LOOP", "1")
    host = capi.run_em(hmm, "hipstr_", **kw)
    monkeypatch.delenv("HIPSTR_EM_HOST_LOOP")
    assert _same(host, want) and _same(got, host)


def test_empty_batch_and_errors(hmm):
    assert all(len(x) == 0 for x in capi.run_em(hmm, "hipstr_", [], [], [0], [], [], [], []))
    with pytest.raises(RuntimeError, match="ascending sample"):
        capi.run_em(hmm, "hipstr_", [4], [2], [0, 3], [1, 0, 1], [0, 4, 4], [0, 0, 0], [0, 0, 0])
