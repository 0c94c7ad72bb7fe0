"""GPU: hipstr_post_extract (genotype calls from the resident posteriors) against the compiled reference's golden vectors and the
oracle.  Tolerance: the streaming / exact log-sum-exps use the device's exp/log, so real-valued outputs are compared with
|d| <= 1e-9 * max(1, |x|) (observed ~1e-13); MAP haplotypes and genotypes must be identical; values that pass through the reference's
FLOAT pair log-sum-exp (GL, GLDIFF, PL, unphased haplotype posterior) may sit one float rounding step (<= 3e-6) away on <= 0.5 % of the
values — and only where that step is OWED: util.assert_genotypes_close recomputes the argument the reference casts to float from the oracle's
posteriors and fails any differing value whose argument is not within 1e-11 (relative) of a float rounding boundary.  test_float_steps_vanish_with_host_libm shows where those steps come from:
with the three exp/log sites evaluated by the host libm (HIPSTR_DEBUG_HOST_LIBM=1) EVERY output is bit-identical to the reference."""
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
def test_golden_fixtures(hmm, oracle, path):
    pb, nv, h2a, exp = util.load_gt_fixture(path)
    got = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a)
    util.assert_genotypes_close(got, exp, TOL, os.path.basename(path), verify=(oracle, pb, nv, h2a))


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
    util.assert_genotypes_close(got, want, TOL, verify=(oracle, pb, [V] * nl, h2a))


def test_outputs_can_be_switched_off_and_errors(hmm):
    pb, nv, h2a, exp = util.load_gt_fixture(FIXTURES[1])
    got = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a, calc_gls=False, calc_pls=False, calc_phased_gls=False)
    assert np.array_equal(got["best_gt"], exp["best_gt"])
    assert np.all(np.abs(got["log_phased_post"] - exp["log_phased_post"]) <= TOL * np.maximum(1, np.abs(exp["log_phased_post"])))
    bad = np.array(h2a).copy(); bad[0] = 10 ** 6
    with pytest.raises(RuntimeError, match="out of range"):
        capi.run_gt_extract(hmm, "hipstr_", pb, nv, bad)


def test_float_steps_vanish_with_host_libm(hmm, oracle, monkeypatch):
    """The widened window of assert_genotypes_close is explained, not assumed: the device's exp / log put ~1e-13 of noise on the
    posteriors, the reference's float pair log-sum-exp (mathops.cpp:86-95) turns that into a float rounding step now and then.  With
    the per-sample log-sum-exp over the diplotypes, the streaming log-sum-exps per genotype and the exact pair log-sum-exp evaluated on
    the host (glibc, the reference's order) on the device's accumulated values, every output — posteriors, GL, GLDIFF, PL, PHASEDGL —
    must equal the compiled reference's golden vectors bit for bit (tol = 0), and the oracle on a 1000-sample locus."""
    monkeypatch.setenv("HIPSTR_DEBUG_HOST_LIBM", "1")
    for path in FIXTURES:
        pb, nv, h2a, exp = util.load_gt_fixture(path)
        got = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a)
        util.assert_genotypes_close(got, exp, 0, "host libm, " + os.path.basename(path))
    # the shape where the steps were counted (15 of 3000 samples at S = 1000): three loci x 1000 samples x 32 haplotypes, one haploid
    rng = np.random.default_rng(42)
    nl, A, S, V = 3, 32, 1000, 8
    counts = rng.integers(3, 9, size=nl * S)
    lab = np.concatenate([np.repeat(np.arange(S), counts[l * S:(l + 1) * S]) for l in range(nl)])
    read_off = np.concatenate([[0], np.cumsum([counts[l * S:(l + 1) * S].sum() for l in range(nl)])]).astype(np.int32)
    n = int(read_off[-1])
    pb = capi.PostBatch([A] * nl, [S] * nl, read_off, lab, -rng.random(n) * 3, -rng.random(n) * 0.05, np.ones(n, np.int32),
                        (-rng.random(n * A) * 40), [0, 1, 0])
    h2a = np.tile((np.arange(A) // 2) % V, nl)
    want = capi.run_gt_extract(oracle, "oracle_", pb, [V] * nl, h2a)
    got = capi.run_gt_extract(hmm, "hipstr_", pb, [V] * nl, h2a)
    util.assert_genotypes_close(got, want, 0, "host libm, 3 x 1000 samples")
    monkeypatch.delenv("HIPSTR_DEBUG_HOST_LIBM")
    steps = util.assert_genotypes_close(capi.run_gt_extract(hmm, "hipstr_", pb, [V] * nl, h2a), want, TOL, "device exp/log, 3 x 1000 samples",
                                        verify=(oracle, pb, [V] * nl, h2a))
    print("device exp/log: %d of %d float-LSE values one float step away; host libm: 0" % (steps[0], steps[1]))
