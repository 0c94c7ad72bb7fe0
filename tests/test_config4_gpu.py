"""GPU: the per-locus shape of BASELINE configs[3] (1000 samples per locus) — posteriors, genotype calls and the stutter EM at
S = 1000, and at A = 128 candidate haplotypes with >= 20 reads per sample — against the oracle.  hs_posterior_kernel runs one
workgroup per (locus, sample): these are the fan-outs (1000-3000 workgroups per locus, 16 K diplotypes per workgroup) the
smaller tests never reach.  Posteriors and genotype calls: bit for bit (util.assert_arrays_exact / assert_genotypes_exact, round 5); EM as in test_em_gpu.py."""
import numpy as np
import pytest

from hipstr_amd import capi
from em_cases import em_case
import util

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _post_case(seed, nl, A, S, reads_per_sample, haploid=None):
    rng = np.random.default_rng(seed)
    rps = rng.integers(reads_per_sample[0], reads_per_sample[1] + 1, size=(nl, S))
    read_off = np.concatenate([[0], np.cumsum(rps.sum(axis=1))])
    lab = np.concatenate([np.repeat(np.arange(S), rps[l]) for l in range(nl)])
    n = int(read_off[-1])
    # likelihood rows that look like alignments: a best allele, neighbours a few nats worse, the rest far off
    best = rng.integers(0, A, size=n)
    ll = -np.abs(np.arange(A)[None, :] - best[:, None]) * rng.uniform(0.5, 3.0, size=(n, 1)) - rng.random((n, A))
    w = (rng.random(n) > 0.1).astype(np.int32)       # zero-weight second mates (genotyper.h:44-46)
    p1 = np.where(rng.random(n) < 0.3, -rng.random(n) * 6, 0.0); p2 = np.where(p1 < 0, -rng.random(n) * 0.05, 0.0)
    return capi.PostBatch([A] * nl, [S] * nl, read_off, lab, p1, p2, w, ll.ravel(), haploid)


def _close(a, b):
    big = b < -1e300
    return np.array_equal(a < -1e300, big) and np.all(np.abs(a[~big] - b[~big]) <= TOL * np.maximum(1, np.abs(b[~big])))


def test_posteriors_and_calls_1000_samples_128_haplotypes(hmm, oracle):
    """S = 1000, A = 128, 20-24 reads per sample (21.9 k reads, 16.4 M diplotype posteriors for the one locus)."""
    A, S, V = 128, 1000, 32
    pb = _post_case(41, 1, A, S, (20, 24))
    want = capi.run_posteriors(oracle, "oracle_", pb)
    got = capi.run_posteriors(hmm, "hipstr_", pb) if hasattr(hmm, "hipstr_posteriors") else None
    if got is None:
        post = np.zeros(int(pb.post_off[-1])); tot = np.zeros(S); gt = np.zeros(2 * S, np.int32); lt = np.zeros(1)
        assert hmm.hipstr_post_run(pb.ptr, None, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p),
                                   gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p)) == 0, hmm.hipstr_last_error()
        got = (post, tot, gt.reshape(-1, 2), lt)
    def cr_post():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pb)
    util.assert_arrays_exact(got, want, cr_post, "posteriors S = 1000, A = 128")
    h2a = (np.arange(A) // 2) % V                    # 2 x 32 x 2 flank options around 32 STR alleles
    want_gt = capi.run_gt_extract(oracle, "oracle_", pb, [V], h2a)
    got_gt = capi.run_gt_extract(hmm, "hipstr_", pb, [V], h2a)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_gt_extract(oracle, "oracle_", pb, [V], h2a)
    util.assert_genotypes_exact(got_gt, want_gt, cr, "calls S = 1000, A = 128", verify=(oracle, pb, [V], h2a))


def test_posteriors_and_calls_1000_samples_several_loci(hmm, oracle):
    """Three loci x 1000 samples x 32 haplotypes at 5x (configs[3] at the usual allele count), one haploid."""
    A, S, V = 32, 1000, 8
    pb = _post_case(42, 3, A, S, (3, 8), haploid=[0, 1, 0])
    want = capi.run_posteriors(oracle, "oracle_", pb)
    post = np.zeros(int(pb.post_off[-1])); tot = np.zeros(3 * S); gt = np.zeros(6 * S, np.int32); lt = np.zeros(3)
    assert hmm.hipstr_post_run(pb.ptr, None, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p),
                               gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p)) == 0, hmm.hipstr_last_error()
    def cr_post():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pb)
    util.assert_arrays_exact((post, tot, gt.reshape(-1, 2), lt), want, cr_post, "posteriors 3 x 1000 samples")
    h2a = np.tile((np.arange(A) // 2) % V, 3)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_gt_extract(oracle, "oracle_", pb, [V] * 3, h2a)
    util.assert_genotypes_exact(capi.run_gt_extract(hmm, "hipstr_", pb, [V] * 3, h2a),
                                capi.run_gt_extract(oracle, "oracle_", pb, [V] * 3, h2a), cr, "calls 3 x 1000 samples", verify=(oracle, pb, [V] * 3, h2a))


def test_stutter_em_1000_samples(hmm, oracle):
    """EMStutterGenotyper::train on loci with 1000 samples each (5-8 reads per sample), every locus against the oracle."""
    kw = em_case(43, n_loci=6, samples=(1000, 1000), reads_per_sample=(5, 8), allele_counts=[3, 9, 5, 12, 4, 7])
    got = capi.run_em(hmm, "hipstr_", **kw)
    want = capi.run_em(oracle, "oracle_", **kw)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert np.all(np.abs(got[1] - want[1]) <= 1e-9) and np.all(np.abs(got[3] - want[3]) <= 1e-9 * np.maximum(1, np.abs(want[3])))


def test_no_map_diplotype_is_reported_like_the_reference(hmm, oracle):
    """All-NaN priors: no diplotype exceeds -DBL_MAX, Genotyper::get_optimal_haplotypes leaves (-1,-1) (genotyper.cpp:84)."""
    pb = capi.PostBatch([3], [2], [0, 4], [0, 0, 1, 1], [0] * 4, [0] * 4, [1] * 4, -np.arange(12.0),
                        log_prior=np.concatenate([np.full(9, np.nan), np.zeros(9)]))
    post = np.zeros(18); tot = np.zeros(2); gt = np.zeros(4, np.int32); lt = np.zeros(1)
    assert hmm.hipstr_post_run(pb.ptr, None, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p),
                               gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p)) == 0
    want = capi.run_posteriors(oracle, "oracle_", pb)
    assert np.array_equal(gt.reshape(-1, 2), want[2]) and tuple(gt[:2]) == (-1, -1) and gt[2] >= 0
    out = capi.run_gt_extract(hmm, "hipstr_", pb, [3], [0, 1, 2])
    assert tuple(out["best_gt"][0]) == (-1, -1) and np.isnan(out["log_phased_post"][0]) and out["best_gt"][1][0] >= 0
