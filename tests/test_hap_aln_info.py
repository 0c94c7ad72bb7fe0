"""Haplotype::get_aln_info() (aln_haps_to_ref + adjust_indels, Haplotype.cpp:8-86): the strings stored with the golden traceback
fixtures come from the compiled reference; the oracle (CPU) and hipstr_hap_aln_info (GPU Needleman-Wunsch) must reproduce them."""
import glob
import os

import pytest

from hipstr_amd import capi
import util

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "trace_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[6:-4] for p in FIXTURES])
def test_oracle_matches_golden(oracle, path):
    for b, _, _, h2r, _ in util.load_trace_fixture(path):
        assert capi.hap_aln_info(oracle, "oracle_", b.ptr) == h2r


@pytest.mark.skipif(not os.path.exists(capi.REF_LIB), reason="compiled reference (oracle/_ref) not built")
def test_oracle_matches_compiled_reference_with_flank_indels(oracle):
    """Alternative flanks that differ from the reference flank by an indel next to the repeat: adjust_indels must move it."""
    ref = capi.load_ref()
    lf, rf = "ACGTTGCATGCATGACCTGAGTCCATGACTTGACA", "TTGACCGTAGGCTAGGCTTAACGGATCCGATTAGC"
    lf_alts = [lf[:-3] + "A" + lf[-3:], lf[:-6] + lf[-4:], lf[:20] + "TT" + lf[20:], lf[:-1]]
    b, A = util.simple_locus(lf, ["CA" * 10, "CA" * 12, "CA" * 7, "CA" * 5 + "CT" + "CA" * 4], rf, 2, [(lf[5:] + "CA" * 10 + rf[:20], None, 5, True)],
                             lf_opts=lf_alts, rf_opts=["C" + rf, rf[2:]])
    b = b.finalize()
    assert A == 5 * 4 * 3
    want = capi.ref_hap_aln_info(ref, b.ptr, A)
    assert capi.hap_aln_info(oracle, "oracle_", b.ptr) == want
    assert any(b"I" in w for w in want) and any(b"D" in w for w in want)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[6:-4] for p in FIXTURES])
def test_gpu_matches_golden(hmm, path):
    for b, _, _, h2r, _ in util.load_trace_fixture(path):
        assert capi.hap_aln_info(hmm, "hipstr_", b.ptr) == h2r


@pytest.mark.gpu
def test_gpu_matches_oracle_on_a_multi_locus_batch(hmm, oracle):
    sb = capi.SynthBatch(n_loci=6, reads_per_locus=4, n_str_alleles=8, n_flank_opts=3, seed=91)
    assert capi.hap_aln_info(hmm, "hipstr_", sb.ptr) == capi.hap_aln_info(oracle, "oracle_", sb.ptr)
