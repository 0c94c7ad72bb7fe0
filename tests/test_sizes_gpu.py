"""GPU: the sizes the reference allows and no test reached before round 6 (VERDICT r05 "missing" 3, "weak" 1).

* BASELINE configs[3]'s forward shape — 5000 pooled reads per locus (1000 samples x 5x) x 32 alleles: two loci against the oracle bit for
  bit, and `bench.py --workload c4`'s whole batch (100 loci, 16 M alignments) through the size-independent properties of
  test_north_star_batch_full_size.  Read packing into 256-lane groups and the HIPSTR_WS_GIB chunking had only been checked to 600 reads
  per locus.
* Up to MAX_TOTAL_HAPLOTYPES = 1000 candidate haplotypes per locus (genotyper_bam_processor.h:110, enforced at
  seq_stutter_genotyper.cpp:610-614): forward + traceback on 5 x 40 x 5 = 1000 and 4 x 60 x 4 = 960 haplotypes (the latter also as golden
  fixtures of the compiled reference: align_/trace_many_haplotypes.npz, run by test_hmm_gpu / test_trace_gpu), posteriors and genotype calls
  with 10^6 diplotypes per sample against the compiled reference's outputs (bigpost_thousand_haplotypes.npz) and the oracle.
* STR periods 1 and 7..9 (stutter_model.h:38 allows 1..9; the generator draws 2..6) with every interruption mode.
"""
import os

import numpy as np
import pytest

from hipstr_amd import capi
from cases import thousand_haplotype_posteriors, BIGPOST_STRIDE
from test_sizes_oracle import load_bigpost
import util

pytestmark = pytest.mark.gpu
C4 = dict(reads_per_locus=5000, n_str_alleles=32, read_len=150, flank_len=60, str_bp=40, seed=20260928)      # bench.py WORKLOADS["c4"]


def test_forward_5000_reads_per_locus_against_the_oracle(hmm, oracle):
    """Loci 0 and 57 of the c4 set, 160 000 alignments each, alone and together."""
    parts = []
    for l in (0, 57):
        one = capi.SynthBatch(n_loci=1, first_locus=l, **C4)
        want, ws = capi.run_align(oracle, "oracle_", one.ptr)
        got, gs = capi.run_align(hmm, "hipstr_hmm_", one.ptr)
        assert np.array_equal(gs, ws) and np.array_equal(got, want), "locus %d: max|diff| %g" % (l, np.max(np.abs(got - want)))
        assert (gs >= 0).sum() > 4000
        parts.append((want, ws))
    two = capi.SynthBatch(n_loci=2, first_locus=56, **C4)           # locus 57 behind another 5000-read locus of the same call
    got, gs = capi.run_align(hmm, "hipstr_hmm_", two.ptr)
    lo = int(two.out_off[1])
    assert np.array_equal(got[lo:], parts[1][0]) and np.array_equal(gs[5000:], parts[1][1])


def test_config4_batch_full_size(hmm, oracle, monkeypatch):
    """The whole `bench.py --workload c4` batch, 100 loci x 5000 reads x 32 alleles = 16 M alignments in one call: finite and <= 0
    (compute_aln_logprob's assert, HapAligner.cpp:229), seeds as the host computes them, a second run identical, four loci spread over the
    batch identical when re-run alone, one of them bit-equal to the oracle; and the same batch with a workspace budget that forces the
    chunked path (HIPSTR_WS_GIB) gives the same bits."""
    NL, P = 100, C4["reads_per_locus"]
    big = capi.SynthBatch(n_loci=NL, **C4)
    dev = hmm.hipstr_hmm_upload(big.ptr); assert dev, hmm.hipstr_last_error()
    runs = []
    for _ in range(2):
        assert hmm.hipstr_hmm_align(dev, None) == 0
        p = np.zeros(big.n_out); s = np.zeros(big.n_reads, np.int32)
        assert hmm.hipstr_hmm_fetch(dev, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0
        runs.append((p, s))
    hmm.hipstr_hmm_free(dev)
    got, seeds = runs[0]
    assert np.array_equal(got, runs[1][0]) and np.array_equal(seeds, runs[1][1])
    assert np.all(np.isfinite(got)) and np.all(got <= 1e-10)
    host_seeds = np.zeros(big.n_reads, np.int32)
    assert hmm.hipstr_calc_seed_bases(big.ptr, host_seeds.ctypes.data_as(capi._i32p)) == 0 and np.array_equal(seeds, host_seeds)
    for n, l in enumerate((3, 38, 71, 99)):
        one = capi.SynthBatch(n_loci=1, first_locus=l, **C4)
        lo, hi = int(big.out_off[l]), int(big.out_off[l + 1])
        alone, s1 = capi.run_align(hmm, "hipstr_hmm_", one.ptr)
        assert np.array_equal(alone, got[lo:hi]) and np.array_equal(s1, seeds[l * P:(l + 1) * P]), "locus %d depends on its batch" % l
        if n == 1:
            want, ws = capi.run_align(oracle, "oracle_", one.ptr)
            assert np.array_equal(want, got[lo:hi]) and np.array_equal(ws, s1), "locus %d differs from the oracle" % l
    monkeypatch.setenv("HIPSTR_WS_GIB", "0.25")
    part = capi.SynthBatch(n_loci=12, first_locus=30, **C4)
    chunked, cs = capi.run_align(hmm, "hipstr_hmm_", part.ptr)
    lo, hi = int(big.out_off[30]), int(big.out_off[42])
    assert np.array_equal(chunked, got[lo:hi]) and np.array_equal(cs, seeds[30 * P:42 * P])


@pytest.mark.parametrize("kw", [
    dict(reads_per_locus=40, n_str_alleles=40, n_flank_opts=5, seed=21),                        # 5 x 40 x 5 = 1000
    dict(reads_per_locus=24, n_str_alleles=60, n_flank_opts=4, seed=22, mask_rate=0.2),        # 4 x 60 x 4 = 960, a fifth masked
    dict(reads_per_locus=16, n_str_alleles=250, n_flank_opts=2, seed=23, str_bp=300, read_len=200, flank_len=50),   # 2 x 250 x 2: a wide STR family
], ids=["5x40x5", "4x60x4_masked", "2x250x2"])
def test_forward_and_traceback_with_a_thousand_haplotypes(hmm, oracle, kw):
    sb = capi.SynthBatch(n_loci=1, **kw)
    A = sb.n_out // sb.n_reads
    assert A >= 900, A
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    assert np.array_equal(gs, ws) and np.array_equal(got, want), np.max(np.abs(got - want))
    rng = np.random.default_rng(kw["seed"])
    rr, aa = [], []
    for r in range(sb.n_reads):
        if ws[r] >= 0:
            for k in list(rng.choice(A, size=4, replace=False)) + [A - 1]:
                rr.append(r); aa.append(int(k))
    h2r = util.synthetic_hap_to_ref(oracle, sb.ptr)
    want_t = capi.run_trace(oracle, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 22)
    got_t = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 22)
    util.assert_traces_equal(got_t, want_t, str(kw))


def test_posteriors_and_calls_with_1000_haplotypes(hmm, oracle):
    """A = 1000 haplotypes = 10^6 diplotypes per sample, three samples (one without reads), 250 variants: against the compiled reference's
    outputs (golden) and the oracle, under the two-level contract of util.assert_arrays_exact / assert_genotypes_exact."""
    kw, nv, h2a = thousand_haplotype_posteriors()
    pb = capi.PostBatch(**kw)
    d, exp = load_bigpost()
    S = 3
    post = np.zeros(int(pb.post_off[-1])); tot = np.zeros(S); gt = np.zeros(2 * S, np.int32); lt = np.zeros(1)
    assert hmm.hipstr_post_run(pb.ptr, None, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p),
                               gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p)) == 0, hmm.hipstr_last_error()
    got = (post, tot, gt.reshape(-1, 2), lt)
    want = capi.run_posteriors(oracle, "oracle_", pb)
    assert np.array_equal(want[0][::BIGPOST_STRIDE], d["expect_post_strided"]) and np.array_equal(want[1], d["expect_total"])      # the checker is the reference's
    def cr_post():
        with capi.oracle_cr_math(oracle):
            return capi.run_posteriors(oracle, "oracle_", pb)
    util.assert_arrays_exact(got, want, cr_post, "posteriors A = 1000")
    got_gt = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a)
    def cr():
        with capi.oracle_cr_math(oracle):
            return capi.run_gt_extract(oracle, "oracle_", pb, nv, h2a)
    util.assert_genotypes_exact(got_gt, exp, cr, "calls A = 1000 (golden)", verify=(oracle, pb, nv, h2a))


@pytest.mark.parametrize("period", [1, 7, 8, 9])
def test_generated_loci_of_periods_1_and_7_to_9(hmm, oracle, monkeypatch, period):
    """stutter_model.h:38 allows periods 1..9; the generator's weights draw 2..6.  HIPSTR_SYNTH_PERIOD forces the others: homopolymer runs
    (period 1: every shift of the block is a repeat shift) and motifs longer than the six-unit artifact span of short alleles, plain and with
    interruptions (every alt allele substituted / inherited from the reference allele), forward and traceback."""
    monkeypatch.setenv("HIPSTR_SYNTH_PERIOD", str(period))
    total = 0
    for imperfect, inherit, kw in [("0.05", "0", dict(n_loci=3, reads_per_locus=40, n_str_alleles=24, str_bp=5 * period + 20)),
                                   ("1.0", "0", dict(n_loci=2, reads_per_locus=30, n_str_alleles=12, str_bp=6 * period)),
                                   ("0.3", "2", dict(n_loci=2, reads_per_locus=30, n_str_alleles=16, str_bp=12 * period, read_len=200, flank_len=50, n_flank_opts=2)),
                                   ("0.0", "0", dict(n_loci=2, reads_per_locus=20, n_str_alleles=40, str_bp=3 * period))]:
        monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", imperfect); monkeypatch.setenv("HIPSTR_SYNTH_INHERIT", inherit)
        sb = capi.SynthBatch(seed=600 + period, **kw)
        assert set(np.ctypeslib.as_array(sb.ptr.contents.period, shape=(kw["n_loci"],))) == {period}
        want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
        assert np.array_equal(gs, ws) and np.array_equal(got, want), (period, imperfect, inherit, kw, np.max(np.abs(got - want)))
        total += got.size
        one = capi.SynthBatch(seed=600 + period, **dict(kw, n_loci=1))
        A = one.n_out // one.n_reads
        _, s1 = capi.run_align(oracle, "oracle_", one.ptr)
        rr = [r for r in range(one.n_reads) if s1[r] >= 0][:12]; aa = [(7 * r + 1) % A for r in rr]
        h2r = util.synthetic_hap_to_ref(oracle, one.ptr)
        util.assert_traces_equal(capi.run_trace(hmm, "hipstr_hmm_", one.ptr, rr, aa, h2r, cap=1 << 21),
                                 capi.run_trace(oracle, "oracle_", one.ptr, rr, aa, h2r, cap=1 << 21), "period %d" % period)
    assert total > 5000


@pytest.mark.parametrize("period,imperfect,n_str", [(9, "1.0", 125), (9, "0.0", 125), (7, "1.0", 160), (6, "1.0", 180), (4, "1.0", 300)],
                         ids=["p9_interrupted", "p9_periodic", "p7_interrupted", "p6_interrupted", "p4_interrupted_1200bp"])
def test_str_alleles_beyond_1024_bp_in_a_wide_family(hmm, oracle, monkeypatch, period, imperfect, n_str):
    """Found by tools/fuzz_align.py "big" in round 6 (a bug since round 2): hs_str_group_kernel / _pw / _rp fetch an allele's block four
    bases per lane of a 256-lane workgroup — 1024 bases — and alleles of 1026 ... 2047 bp that are interrupted, or periodic with a period
    above six (no hs_str_group_kernel_p), were scored from a truncated block: 156 wrong entries of 5000 in this very locus, values like
    2.7e278.  prep.cpp now leaves such blocks to the per-read kernels (layout.h HS_GRP_MAX_BLOCK); the traceback takes them too.  The only earlier test beyond 1024 bp
    (test_limits_gpu.py::test_1500bp_allele) has periodic ACAG blocks, which hs_str_group_kernel_p takes without fetching them."""
    monkeypatch.setenv("HIPSTR_SYNTH_PERIOD", str(period)); monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", imperfect); monkeypatch.setenv("HIPSTR_SYNTH_INHERIT", "0")
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=40, n_str_alleles=n_str, read_len=67, flank_len=95, str_bp=10, seed=564467067)
    nopt = np.ctypeslib.as_array(sb.ptr.contents.blk_nopts, shape=(3,))
    lens = np.diff(np.ctypeslib.as_array(sb.ptr.contents.opt_off, shape=(int(nopt.sum()) + 1,)))[nopt[0]:nopt[0] + nopt[1]]
    assert lens.max() > 1024 and lens.max() <= 2047
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    assert np.array_equal(gs, ws) and np.array_equal(got, want), (int((got != want).sum()), float(np.nanmax(np.abs(got - want))))
    # ... and the traceback against the longest alleles (until round 6 hipstr_hmm_trace refused an STR allele of more than 1024 bp — a limit
    # from round 1 that nothing in the kernels needed: hs_trace_fill keeps blocks of up to 2047 bp in LDS)
    order = np.argsort(lens, kind="stable")
    rr = [r for r in range(sb.n_reads) if ws[r] >= 0][:8]; aa = [int(order[-1 - (i % 3)]) for i in range(len(rr))]       # one flank option each: allele index = STR option
    assert lens[aa].min() > 1024
    h2r = util.synthetic_hap_to_ref(oracle, sb.ptr)
    util.assert_traces_equal(capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 23), capi.run_trace(oracle, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 23), "alleles beyond 1024 bp")
