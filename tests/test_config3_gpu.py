"""GPU: BASELINE configs[2] at its own size — 10 000 loci x 600 reads (100 samples x 6) x 32 alleles, de novo stutter EM + forward HMM +
posteriors + genotype calls, as `bench.py --workload c3` runs it (VERDICT r03 "next round" item 3; the shape had only been tested shrunk).

The oracle cannot do 192 M alignments, so the whole batch goes through the size-independent properties and a strided sample of loci is held
to the oracle:
  * forward scores: finite, <= 0, identical when the resident batch is run twice; ten loci spread over the batch identical when regenerated
    from (seed, index) and run ALONE, and bit-equal to the oracle;
  * stutter EM over all 10 000 loci: identical when run twice; twenty loci (every 500th) trained ALONE give the same iteration counts,
    train() results and parameters bit for bit, and agree with the oracle (iteration counts and results identical, parameters <= 1e-9);
  * posteriors + genotype calls of the strided loci: in the batch == alone (bits), and against the oracle under the contract of
    tests/util.py::assert_genotypes_close (float steps only where owed)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from hipstr_amd import capi
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NL, P, A, S, SEED = 10000, 600, 32, 100, 20260928


def _post_batch(A_l, nl):
    lab = np.tile(np.repeat(np.arange(S), P // S), nl).astype(np.int32)
    return capi.PostBatch(A_l, np.full(nl, S, np.int32), np.arange(nl + 1, dtype=np.int32) * P, lab, np.zeros(nl * P), np.zeros(nl * P),
                          np.ones(nl * P, np.int32), None)


def _cut_em(kw, loci):
    """the EM input of a few loci of a batch, as a batch of their own"""
    ro = np.asarray(kw["read_off"])
    out = {k: np.asarray(kw[k])[loci] for k in ("period", "n_samples", "haploid")}
    out["read_off"] = np.concatenate([[0], np.cumsum([ro[l + 1] - ro[l] for l in loci])]).astype(np.int32)
    for k in ("sample_label", "num_bps", "log_p1", "log_p2"):
        out[k] = np.concatenate([np.asarray(kw[k])[ro[l]:ro[l + 1]] for l in loci])
    return out


def test_config3_full_size(hmm, oracle):
    import bench
    big = capi.SynthBatch(n_loci=NL, reads_per_locus=P, n_str_alleles=A, seed=SEED)
    A_l = np.diff(np.ctypeslib.as_array(big.ptr.contents.hap_off, shape=(NL + 1,)))
    # ---- forward HMM, twice on the resident batch
    dev = hmm.hipstr_hmm_upload(big.ptr); assert dev, hmm.hipstr_last_error()
    runs = []
    for _ in range(2):
        assert hmm.hipstr_hmm_align(dev, None) == 0
        p = np.zeros(big.n_out); s = np.zeros(big.n_reads, np.int32)
        assert hmm.hipstr_hmm_fetch(dev, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0
        runs.append((p, s))
    got, seeds = runs[0]
    assert np.array_equal(got, runs[1][0]) and np.array_equal(seeds, runs[1][1])
    assert np.all(np.isfinite(got)) and np.all(got <= 1e-10)
    del runs
    # ---- posteriors of every (locus, sample) on the resident likelihoods
    pb = _post_batch(A_l, NL)
    pd = hmm.hipstr_post_upload(pb.ptr, hmm.hipstr_hmm_dev_aln_probs(dev)); assert pd, hmm.hipstr_last_error()
    assert hmm.hipstr_post_launch(pd, None) == 0
    post = np.zeros(int(pb.post_off[-1])); tot = np.zeros(NL * S); gt = np.zeros(2 * NL * S, np.int32); lt = np.zeros(NL)
    assert hmm.hipstr_post_fetch(pd, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p), gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p)) == 0
    hmm.hipstr_post_free(pd); hmm.hipstr_hmm_free(dev)
    assert np.all(np.isfinite(tot)) and np.all(tot <= 1e-9) and np.all(gt >= 0) and np.all(post <= 1e-9)
    gt = gt.reshape(-1, 2)
    # ---- ten loci alone == in the batch == oracle (forward scores; posteriors and genotype calls)
    for l in range(7, NL, 1000):
        one = capi.SynthBatch(n_loci=1, reads_per_locus=P, n_str_alleles=A, seed=SEED, first_locus=l)
        lo, hi = int(big.out_off[l]), int(big.out_off[l + 1])
        alone, s1 = capi.run_align(hmm, "hipstr_hmm_", one.ptr)
        assert np.array_equal(alone, got[lo:hi]) and np.array_equal(s1, seeds[l * P:(l + 1) * P]), "locus %d depends on its batch" % l
        want, ws = capi.run_align(oracle, "oracle_", one.ptr)
        assert np.array_equal(want, got[lo:hi]) and np.array_equal(ws, s1), "locus %d differs from the oracle" % l
        Al = int(A_l[l])
        pb1 = capi.PostBatch([Al], [S], [0, P], np.repeat(np.arange(S), P // S), np.zeros(P), np.zeros(P), np.ones(P, np.int32), want.copy())
        p1 = np.zeros(int(pb1.post_off[-1])); t1 = np.zeros(S); g1 = np.zeros(2 * S, np.int32); l1 = np.zeros(1)
        assert hmm.hipstr_post_run(pb1.ptr, None, p1.ctypes.data_as(capi._f64p), t1.ctypes.data_as(capi._f64p), g1.ctypes.data_as(capi._i32p), l1.ctypes.data_as(capi._f64p)) == 0
        plo = int(pb.post_off[l])
        assert np.array_equal(p1, post[plo:plo + p1.size]) and np.array_equal(t1, tot[l * S:(l + 1) * S]) and np.array_equal(g1.reshape(-1, 2), gt[l * S:(l + 1) * S])
        wp = capi.run_posteriors(oracle, "oracle_", pb1)
        def cr_post():
            with capi.oracle_cr_math(oracle):
                return capi.run_posteriors(oracle, "oracle_", pb1)
        util.assert_arrays_exact((p1, t1, g1.reshape(-1, 2)), wp[:3], cr_post, "configs[2] locus %d posteriors" % l)
        h2a = np.arange(Al, dtype=np.int32)
        def cr():
            with capi.oracle_cr_math(oracle):
                return capi.run_gt_extract(oracle, "oracle_", pb1, [Al], h2a)
        util.assert_genotypes_exact(capi.run_gt_extract(hmm, "hipstr_", pb1, [Al], h2a), capi.run_gt_extract(oracle, "oracle_", pb1, [Al], h2a), cr,
                                    "configs[2] locus %d" % l, verify=(oracle, pb1, [Al], h2a))
    # ---- stutter EM: all loci in lock step, twice; every 500th locus alone and against the oracle
    kw = bench.c3_em_inputs(big, NL, P, S)
    em = capi.run_em(hmm, "hipstr_", **kw)
    again = capi.run_em(hmm, "hipstr_", **kw)
    assert all(np.array_equal(a, b) for a, b in zip(em, again))
    assert np.all(np.isfinite(em[1])) and np.all(em[2] >= 1)
    sample = list(range(3, NL, 500))
    cut = _cut_em(kw, sample)
    alone = capi.run_em(hmm, "hipstr_", **cut)
    assert all(np.array_equal(np.asarray(a)[sample], b) for a, b in zip(em, alone)), "a locus' training depends on its batch"
    want = capi.run_em(oracle, "oracle_", **cut)
    assert np.array_equal(alone[0], want[0]) and np.array_equal(alone[2], want[2]), "EM iteration counts / train() results differ from the oracle"
    assert np.all(np.abs(alone[1] - want[1]) <= 1e-9) and np.all(np.abs(alone[3] - want[3]) <= 1e-9 * np.maximum(1, np.abs(want[3])))
