"""GPU: hipstr_post_extract (genotype calls from the resident posteriors) against the compiled reference's golden vectors and the
oracle.  Contract (round 5, util.assert_genotypes_exact): EVERY output — MAP haplotypes and genotypes, posteriors, Q, GL, GLDIFF, PL,
PHASEDGL — equals the reference bit for bit (tolerance 0): the three exp / log sites of the posterior and genotype kernels are evaluated
with correctly rounded functions (hipstr_amd/csrc/cr_math.h) in the reference's summation order.  The one way left to differ is an
argument on which the HOST's libm is not correctly rounded (glibc: ~8 in 10^4 exp arguments, tests/test_cr_math.py); such a case must be
explained completely by the second level: device == the oracle evaluated with the same correctly rounded functions, bit for bit, and that
within the round-4 rule of the host-libm reference (1e-9; a float step of GL only where the reference's cast sits on a rounding boundary)."""
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
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_gt_extract(oracle, "oracle_", pb, nv, h2a)
    util.assert_genotypes_exact(got, exp, cr, os.path.basename(path), verify=(oracle, pb, nv, h2a))


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
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_gt_extract(oracle, "oracle_", pb, [V] * nl, h2a)
    util.assert_genotypes_exact(got, want, cr, "north-star shape", verify=(oracle, pb, [V] * nl, h2a))


def test_outputs_can_be_switched_off_and_errors(hmm):
    pb, nv, h2a, exp = util.load_gt_fixture(FIXTURES[1])
    got = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a, calc_gls=False, calc_pls=False, calc_phased_gls=False)
    assert np.array_equal(got["best_gt"], exp["best_gt"])
    assert np.all(np.abs(got["log_phased_post"] - exp["log_phased_post"]) <= TOL * np.maximum(1, np.abs(exp["log_phased_post"])))
    bad = np.array(h2a).copy(); bad[0] = 10 ** 6
    with pytest.raises(RuntimeError, match="out of range"):
        capi.run_gt_extract(hmm, "hipstr_", pb, nv, bad)


def test_bit_identity_at_1000_samples_with_and_without_host_libm(hmm, oracle, monkeypatch):
    """Three loci x 1000 samples x 32 haplotypes (one haploid): 86 000 GL / GLDIFF / unphased-posterior values.  Round 4 counted 165 of
    them one float step from the reference with the device's own exp / log and 0 with the three exp / log sites evaluated by the host
    (HIPSTR_DEBUG_HOST_LIBM=1).  Both ways must now be bit-identical to the oracle (level 1 of util.assert_genotypes_exact, or level 2 if
    the host libm misrounds an argument of this case), and the golden fixtures with the debug switch as before."""
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
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_gt_extract(oracle, "oracle_", pb, [V] * nl, h2a)
    level1 = util.assert_genotypes_exact(capi.run_gt_extract(hmm, "hipstr_", pb, [V] * nl, h2a), want, cr, "3 x 1000 samples", verify=(oracle, pb, [V] * nl, h2a))
    print("3 x 1000 samples, device (correctly rounded exp/log): %s" % ("bit-identical to the host-libm oracle" if level1 else "bit-identical to the correctly rounded oracle; the host libm misrounds an argument of this case"))
    monkeypatch.setenv("HIPSTR_DEBUG_HOST_LIBM", "1")
    for path in FIXTURES:
        pbf, nv, h2f, exp = util.load_gt_fixture(path)
        util.assert_genotypes_close(capi.run_gt_extract(hmm, "hipstr_", pbf, nv, h2f), exp, 0, "host libm, " + os.path.basename(path))
    util.assert_genotypes_close(capi.run_gt_extract(hmm, "hipstr_", pb, [V] * nl, h2a), want, 0, "host libm, 3 x 1000 samples")
