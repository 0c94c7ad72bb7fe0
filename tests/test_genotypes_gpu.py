"""GPU: hipstr_post_extract (genotype calls from the resident posteriors) against the compiled reference's golden vectors and the
oracle.  Tolerance: the streaming / exact log-sum-exps use the device's exp/log, so real-valued outputs are compared with
|d| <= 1e-9 * max(1, |x|) (observed ~1e-13); MAP haplotypes and genotypes must be identical; a PL (truncated integer) may differ
by one only where -10*(GL - maxGL) sits within 1e-6 of an integer."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "gt_*.npz")))
TOL = 1e-9


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_golden_fixtures(hmm, path):
    pb, nv, h2a, exp = util.load_gt_fixture(path)
    got = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a)
    util.assert_genotypes_close(got, exp, TOL, os.path.basename(path))


def test_north_star_shape_against_oracle(hmm, oracle):
    """32 haplotypes = 2 x 8 x 2 (flank options around 8 STR alleles), 100 samples: the C3 shape of BASELINE configs[2]."""
    rng = np.random.default_rng(11)
    nl, A, S, V = 3, 32, 100, 8
    R = S * 6; n = nl * R
    kw = dict(n_alleles=[A] * nl, n_samples=[S] * nl, read_off=np.arange(nl + 1) * R, sample_label=np.tile(np.repeat(np.arange(S), 6), nl),
              log_p1=-rng.random(n), log_p2=-rng.random(n), read_weight=np.ones(n, np.int32), log_aln_probs=-rng.random(n * A) * 40,
              haploid=[0, 1, 0])
    pb = capi.PostBatch(**kw)
    h2a = np.tile((np.arange(A) // 2) % V, nl)
    want = capi.run_gt_extract(oracle, "oracle_", pb, [V] * nl, h2a)
    got = capi.run_gt_extract(hmm, "hipstr_", pb, [V] * nl, h2a)
    util.assert_genotypes_close(got, want, TOL)


def test_outputs_can_be_switched_off_and_errors(hmm):
    pb, nv, h2a, exp = util.load_gt_fixture(FIXTURES[1])
    got = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a, calc_gls=False, calc_pls=False, calc_phased_gls=False)
    assert np.array_equal(got["best_gt"], exp["best_gt"])
    assert np.all(np.abs(got["log_phased_post"] - exp["log_phased_post"]) <= TOL * np.maximum(1, np.abs(exp["log_phased_post"])))
    bad = np.array(h2a).copy(); bad[0] = 10 ** 6
    with pytest.raises(RuntimeError, match="out of range"):
        capi.run_gt_extract(hmm, "hipstr_", pb, nv, bad)
