"""CPU: the oracle against the compiled reference itself on fresh seeded inputs (skipped where
oracle/_ref/libhipstr_ref.so has not been built — it is built wherever /root/reference is mounted and
travels to the GPU box as a prebuilt artefact)."""
import numpy as np
import pytest

from hipstr_amd import capi

pytestmark = pytest.mark.skipif(not capi.have_ref(), reason="compiled reference not available")


@pytest.fixture(scope="module")
def ref():
    return capi.load_ref()


@pytest.mark.parametrize("kw", [
    dict(n_loci=6, reads_per_locus=12, n_str_alleles=5, seed=101),
    dict(n_loci=4, reads_per_locus=10, n_str_alleles=7, n_flank_opts=2, seed=102),
    dict(n_loci=4, reads_per_locus=10, n_str_alleles=4, n_flank_opts=3, seed=103, mask_rate=0.3),
    dict(n_loci=2, reads_per_locus=6, n_str_alleles=12, read_len=250, flank_len=110, str_bp=100, seed=104),
    dict(n_loci=5, reads_per_locus=10, n_str_alleles=6, read_len=90, flank_len=35, str_bp=24, seed=105),
])
def test_align_matches_reference(oracle, ref, kw):
    sb = capi.SynthBatch(**kw)
    pr, sr = capi.run_align(ref, "ref_", sb.ptr, fill=-1.5)
    po, so = capi.run_align(oracle, "oracle_", sb.ptr, fill=-1.5)
    assert np.array_equal(sr, so)
    assert np.array_equal(pr, po)


def test_gray_code_order(oracle, ref):
    """Allele index <-> per-block option: Haplotype::next() visit order for a [2,3,2] haplotype."""
    import ctypes as C
    from util import simple_locus
    b, A = simple_locus("ACGTACGTAC", ["GA" * 4, "GA" * 5, "GA" * 3], "TTGCATGCAA", 2, [], lf_opts=["ACGTACGTAA"], rf_opts=["ATGCATGCAA"])
    b.finalize()
    buf = C.create_string_buffer(4096); lens = np.zeros(A, np.int32)
    assert ref.ref_hap_sequences(b.ptr, 0, buf, 4096, lens.ctypes.data_as(capi._i32p)) == 0
    raw = buf.raw; pos = 0
    lfs = ["ACGTACGTAC", "ACGTACGTAA"]; strs = ["GA" * 4, "GA" * 5, "GA" * 3]; rfs = ["TTGCATGCAA", "ATGCATGCAA"]
    want_digits = ["000", "100", "110", "010", "020", "120", "121", "021", "011", "111", "101", "001"]
    for k in range(A):
        seq = raw[pos:pos + lens[k]].decode(); pos += lens[k]
        opts = np.zeros(3, np.int32)
        assert oracle.oracle_allele_options(np.array([2, 3, 2], np.int32).ctypes.data_as(capi._i32p), k, opts.ctypes.data_as(capi._i32p)) == 0
        assert "".join(map(str, opts)) == want_digits[k]
        assert seq == lfs[opts[0]] + strs[opts[1]] + rfs[opts[2]]


def test_posteriors_match_reference(oracle, ref):
    rng = np.random.default_rng(5)
    for _ in range(3):
        nl = 4; A = rng.integers(1, 12, nl); S = rng.integers(1, 5, nl); R = [int(rng.integers(s, 6 * s + 1)) for s in S]
        ro = np.concatenate([[0], np.cumsum(R)]); lab = np.concatenate([np.sort(rng.integers(0, s, r)) for s, r in zip(S, R)])
        n = int(ro[-1])
        pb = capi.PostBatch(A, S, ro, lab, -rng.random(n), -rng.random(n) * 4, (rng.random(n) < 0.9).astype(int),
                            np.concatenate([-rng.random(r * a) * 30 for r, a in zip(R, A)]), haploid=(rng.random(nl) < 0.3))
        a = capi.run_posteriors(ref, "ref_", pb); b = capi.run_posteriors(oracle, "oracle_", pb)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_custom_priors_match_reference(oracle, ref):
    """The virtual init_log_sample_priors hook (genotyper.h:69; EMStutterGenotyper overrides it): arbitrary prior arrays."""
    rng = np.random.default_rng(8)
    A = np.array([3, 5]); S = np.array([2, 3]); R = [7, 9]
    ro = np.concatenate([[0], np.cumsum(R)]); lab = np.concatenate([np.sort(rng.integers(0, s, r)) for s, r in zip(S, R)])
    n = int(ro[-1]); prior = -rng.random(int((S * A * A).sum())) * 5
    pb = capi.PostBatch(A, S, ro, lab, -rng.random(n), -rng.random(n), np.ones(n, int), np.concatenate([-rng.random(r * a) * 20 for r, a in zip(R, A)]),
                        log_prior=prior)
    a = capi.run_posteriors(ref, "ref_", pb); b = capi.run_posteriors(oracle, "oracle_", pb)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
