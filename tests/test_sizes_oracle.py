"""CPU: the oracle at the sizes round 6 added fixtures for — 1000 candidate haplotypes per locus (MAX_TOTAL_HAPLOTYPES,
genotyper_bam_processor.h:110; enforced at seq_stutter_genotyper.cpp:610-614) in the posterior / genotype-call stage — against outputs of
the compiled reference (tests/golden/bigpost_thousand_haplotypes.npz, written by make_golden.py's "sizes" section: a strided sample of the
3 x 10^6 posteriors, every other output whole).  The forward and traceback fixtures of the 960-haplotype locus (align_many_haplotypes,
trace_many_haplotypes) are picked up by test_oracle_golden.py / test_trace_oracle.py like every other fixture."""
import os

import numpy as np

from hipstr_amd import capi
from cases import thousand_haplotype_posteriors, BIGPOST_STRIDE
import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bigpost_thousand_haplotypes.npz")


def load_bigpost():
    d = np.load(GOLD)
    exp = {k: d["expect_" + k] for k in ("best_hap", "best_gt", "log_phased_post", "log_unphased_post", "hap_log_phased_post",
                                         "hap_log_unphased_post", "gl_diff")}
    for k in ("gls", "pls", "phased_gls"):
        exp[k] = np.split(d["expect_" + k], np.cumsum(d["expect_" + k + "_len"])[:-1])
    return d, exp


def test_oracle_posteriors_and_calls_with_1000_haplotypes(oracle):
    kw, nv, h2a = thousand_haplotype_posteriors()
    pb = capi.PostBatch(**kw)
    d, exp = load_bigpost()
    post, tot, gt, ltot = capi.run_posteriors(oracle, "oracle_", pb)
    assert post.size == 3 * 1000 * 1000
    assert np.array_equal(post[::BIGPOST_STRIDE], d["expect_post_strided"]) and post.max() == d["expect_post_max"][0]
    assert np.array_equal(tot, d["expect_total"]) and np.array_equal(gt, d["expect_gt"]) and np.array_equal(ltot, d["expect_locus_total"])
    got = capi.run_gt_extract(oracle, "oracle_", pb, nv, h2a)
    util.assert_genotypes_close(got, exp, 0, "1000 haplotypes")
    assert len(got["gls"][0]) == 250 * 251 // 2
