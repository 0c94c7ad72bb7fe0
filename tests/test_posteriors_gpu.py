"""GPU: diplotype posteriors through the C-ABI against the compiled reference's golden fixtures and the oracle.

Contract (round 5): bit for bit.  The per-read accumulation always was; the final exact log-sum-exp over the A^2 diplotypes now sums
correctly rounded exponentials (cr_math.h) in the reference's index order, so posteriors and totals equal the host's — unless the host's
libm is not correctly rounded on an argument of the case, which util.assert_arrays_exact then has to explain completely (device == the
oracle with the same correctly rounded functions, bit for bit; that within 1e-9 of the host-libm reference)."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
import util

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POST = sorted(glob.glob(os.path.join(GOLD, "post_*.npz")))
TOL = 1e-9


def _run(hmm, pb, dev_ll=None):
    S = int(pb.samp_off[-1])
    post = np.zeros(max(int(pb.post_off[-1]), 1)); tot = np.zeros(max(S, 1)); gt = np.zeros(max(2 * S, 2), np.int32)
    ltot = np.zeros(max(pb.struct.n_loci, 1))
    rc = hmm.hipstr_post_run(pb.ptr, dev_ll, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p),
                             gt.ctypes.data_as(capi._i32p), ltot.ctypes.data_as(capi._f64p))
    assert rc == 0, hmm.hipstr_last_error()
    return post[:int(pb.post_off[-1])], tot[:S], gt[:2 * S].reshape(-1, 2), ltot[:pb.struct.n_loci]


def _finite_close(a, b):
    big = (b < -1e300)
    return np.array_equal(a < -1e300, big) and np.all(np.abs(a[~big] - b[~big]) <= TOL * np.maximum(1, np.abs(b[~big])))


@pytest.mark.parametrize("path", POST, ids=[os.path.basename(p)[5:-4] for p in POST])
def test_golden_fixtures(hmm, oracle, path):
    d = np.load(path)
    pb = capi.PostBatch(d["n_alleles"], d["n_samples"], d["read_off"], d["sample_label"], d["log_p1"], d["log_p2"], d["read_weight"],
                        d["log_aln_probs"], d["haploid"])
    post, tot, gt, ltot = _run(hmm, pb)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pb)
    util.assert_arrays_exact((post, tot, gt, ltot), (d["expect_post"], d["expect_total"], d["expect_gt"], d["expect_locus_total"]), cr, os.path.basename(path))


def test_chained_from_device_alignments(hmm, oracle):
    """align -> posteriors without a host round trip of the likelihood matrix; checked against oracle -> oracle."""
    sb = capi.SynthBatch(n_loci=6, reads_per_locus=40, n_str_alleles=12, seed=77)
    dev = hmm.hipstr_hmm_upload(sb.ptr); assert dev
    assert hmm.hipstr_hmm_align(dev, None) == 0
    p = np.zeros(sb.n_out); s = np.zeros(sb.n_reads, np.int32)
    assert hmm.hipstr_hmm_fetch(dev, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0
    nl, R = 6, 40
    A = np.diff(np.ctypeslib.as_array(sb.ptr.contents.hap_off, shape=(nl + 1,)))
    S = np.full(nl, 4); lab = np.tile(np.repeat(np.arange(4), R // 4), nl)
    rng = np.random.default_rng(3); n = nl * R
    kw = dict(n_alleles=A, n_samples=S, read_off=np.arange(nl + 1) * R, sample_label=lab, log_p1=-rng.random(n), log_p2=-rng.random(n),
              read_weight=np.ones(n, np.int32))
    pb_dev = capi.PostBatch(log_aln_probs=None, **kw)
    got = _run(hmm, pb_dev, dev_ll=hmm.hipstr_hmm_dev_aln_probs(dev))
    hmm.hipstr_hmm_free(dev)
    want_ll, _ = capi.run_align(oracle, "oracle_", sb.ptr)
    assert np.array_equal(p, want_ll)
    want = capi.run_posteriors(oracle, "oracle_", capi.PostBatch(log_aln_probs=want_ll, **kw))
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", capi.PostBatch(log_aln_probs=want_ll, **kw))
    util.assert_arrays_exact(got, want, cr, "chained posteriors")
    # posteriors are normalised: sum over diplotypes of exp(log posterior) == 1
    off = 0
    for a in A:
        blk = got[0][off:off + 4 * a * a].reshape(4, -1); off += 4 * a * a
        assert np.allclose(np.exp(blk).sum(axis=1), 1.0, atol=1e-9)


def test_large_allele_count(hmm, oracle):
    """A = 128 (BASELINE configs[4]) with 3 samples."""
    rng = np.random.default_rng(11)
    A, S, R = 128, 3, 60
    pb = capi.PostBatch([A], [S], [0, R], np.repeat(np.arange(S), R // S), -rng.random(R), -rng.random(R), np.ones(R, np.int32),
                        -rng.random(R * A) * 60)
    got = _run(hmm, pb); want = capi.run_posteriors(oracle, "oracle_", pb)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pb)
    util.assert_arrays_exact(got, want, cr, "A = 128")


def test_ungrouped_samples_rejected(hmm):
    pb = capi.PostBatch([2], [2], [0, 3], [0, 1, 0], [0, 0, 0], [0, 0, 0], [1, 1, 1], -np.ones(6))
    post = np.zeros(8); tot = np.zeros(2); gt = np.zeros(4, np.int32); lt = np.zeros(1)
    assert hmm.hipstr_post_run(pb.ptr, None, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p),
                               gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p)) != 0


def test_custom_prior_array(hmm, oracle):
    """Population priors supplied by the caller (what EMStutterGenotyper's init_log_sample_priors override produces)."""
    rng = np.random.default_rng(21)
    A = np.array([4, 9]); S = np.array([3, 2]); R = [12, 10]
    ro = np.concatenate([[0], np.cumsum(R)]); lab = np.concatenate([np.sort(rng.integers(0, s, r)) for s, r in zip(S, R)])
    n = int(ro[-1]); prior = -rng.random(int((S * A * A).sum())) * 6
    pb = capi.PostBatch(A, S, ro, lab, -rng.random(n), -rng.random(n), np.ones(n, int), np.concatenate([-rng.random(r * a) * 25 for r, a in zip(R, A)]),
                        log_prior=prior)
    got = _run(hmm, pb); want = capi.run_posteriors(oracle, "oracle_", pb)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pb)
    util.assert_arrays_exact(got, want, cr, "custom priors")


def test_launch_on_a_stream_of_the_callers(hmm):
    """include/hipstr_hmm.h allows hipstr_post_launch on a stream of the caller's.  A small run sends its inputs and argument block
    asynchronously on the run's own stream; the kernel on the other stream must wait for that copy (it read d_args before the copy
    had landed in round 3: ADVICE r03).  Many small runs back to back, each launched on a fresh non-blocking stream right after its
    upload, against the same runs on the default path."""
    import ctypes as C
    d = np.load(POST[0])
    kw = dict(n_alleles=d["n_alleles"], n_samples=d["n_samples"], read_off=d["read_off"], sample_label=d["sample_label"], log_p1=d["log_p1"],
              log_p2=d["log_p2"], read_weight=d["read_weight"], log_aln_probs=d["log_aln_probs"], haploid=d["haploid"])
    pb = capi.PostBatch(kw["n_alleles"], kw["n_samples"], kw["read_off"], kw["sample_label"], kw["log_p1"], kw["log_p2"], kw["read_weight"],
                        kw["log_aln_probs"], kw["haploid"])
    want = _run(hmm, pb)
    S = int(pb.samp_off[-1])
    streams = []
    for _ in range(4):
        h = hmm.hipstr_debug_stream_create()          # (made by the library's HIP runtime: a second runtime in the process — torch's, or
        assert h, hmm.hipstr_last_error()             #  libamdhip64 loaded by name — does not see the device)
        streams.append(C.c_void_p(h))
    for it in range(40):
        st = streams[it % 4]
        pd = hmm.hipstr_post_upload(pb.ptr, None); assert pd, hmm.hipstr_last_error()
        assert hmm.hipstr_post_launch(pd, st) == 0, hmm.hipstr_last_error()
        post = np.zeros(max(int(pb.post_off[-1]), 1)); tot = np.zeros(max(S, 1)); gt = np.zeros(max(2 * S, 2), np.int32); ltot = np.zeros(max(pb.struct.n_loci, 1))
        assert hmm.hipstr_post_fetch(pd, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p), gt.ctypes.data_as(capi._i32p),
                                     ltot.ctypes.data_as(capi._f64p)) == 0, hmm.hipstr_last_error()
        hmm.hipstr_post_free(pd)
        assert np.array_equal(post[:int(pb.post_off[-1])], want[0]) and np.array_equal(tot[:S], want[1]) and np.array_equal(gt[:2 * S].reshape(-1, 2), want[2])
    for h in streams:
        hmm.hipstr_debug_stream_destroy(h)


def test_split_accumulation_is_bit_identical(hmm, oracle):
    """Few (locus, sample) units with many diplotypes take hs_posterior_accumulate_kernel (several workgroups per unit) +
    hs_posterior_finish_kernel; many units take the one-launch kernel.  The same three loci — one sample each, 96 haplotypes, 50 reads —
    alone (split) and in front of 1100 small single-sample loci (not split) must give the same bits, and agree with the oracle."""
    rng = np.random.default_rng(8)
    A_big, R_big, n_big = 96, 50, 3
    A_small, R_small, n_small = 4, 6, 1100
    def case(n_tail):
        A = [A_big] * n_big + [A_small] * n_tail; R = [R_big] * n_big + [R_small] * n_tail
        off = np.concatenate([[0], np.cumsum(R)]).astype(np.int32); n = int(off[-1])
        return dict(n_alleles=A, n_samples=[1] * len(A), read_off=off, sample_label=np.zeros(n, np.int32)), n, int(np.dot(A, R))
    r0 = np.random.default_rng(9)
    head_n = n_big * R_big; head_ll = -r0.random(head_n * A_big) * 30; head_p1 = -r0.random(head_n); head_p2 = -r0.random(head_n)
    outs = []
    for n_tail in (0, n_small):
        kw, n, nll = case(n_tail)
        p1 = np.concatenate([head_p1, -rng.random(n - head_n)]); p2 = np.concatenate([head_p2, -rng.random(n - head_n)])
        ll = np.concatenate([head_ll, -rng.random(nll - head_ll.size) * 30])
        pb = capi.PostBatch(log_p1=p1, log_p2=p2, read_weight=np.ones(n, np.int32), log_aln_probs=ll, **kw)
        outs.append((_run(hmm, pb), pb))
    (a, pba), (b, pbb) = outs
    npost = n_big * A_big * A_big
    assert np.array_equal(a[0][:npost], b[0][:npost]) and np.array_equal(a[1][:n_big], b[1][:n_big]) and np.array_equal(a[2][:n_big], b[2][:n_big])
    want = capi.run_posteriors(oracle, "oracle_", pba)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pba)
    util.assert_arrays_exact(a, want, cr, "split accumulation")
