"""CPU: the traceback restatement (oracle_trace) against the golden vectors made by the compiled reference
(HapAligner::trace_optimal_aln + stitch_alignment_trace), and against the compiled reference itself where it exists."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
import util

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "trace_*.npz")))


def test_fixtures_present():
    assert len(FIXTURES) >= 10


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[6:-4] for p in FIXTURES])
def test_oracle_trace_matches_golden(path):
    ora = capi.load_oracle()
    n = 0
    for b, rr, aa, h2r, exp in util.load_trace_fixture(path):
        got = capi.run_trace(ora, "oracle_", b.ptr, rr, aa, h2r, cap=1 << 20)
        util.assert_traces_equal(got, exp, os.path.basename(path))
        n += len(exp)
    assert n > 0


def test_golden_covers_the_interesting_outcomes():
    """The fixtures must exercise stutter artifacts of both signs, flank indels, SNPs, soft clips and reads that never
    enter the STR block — otherwise a green comparison proves little."""
    seen = dict(stutter_pos=0, stutter_neg=0, no_str=0, indel=0, snp=0, clip=0)
    for path in FIXTURES:
        for _, _, _, _, exp in util.load_trace_fixture(path):
            for e in exp:
                seen["stutter_pos"] += e["stutter_size"] > 0 and e["stutter_size"] != -100000
                seen["stutter_neg"] += -100000 < e["stutter_size"] < 0
                seen["no_str"] += e["stutter_size"] == -100000
                seen["indel"] += len(e["indels"]) > 0
                seen["snp"] += len(e["snps"]) > 0
                seen["clip"] += "S" in e["hap_aln"]
    assert all(v > 0 for v in seen.values()), seen


@pytest.mark.skipif(not os.path.exists(capi.REF_LIB), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("kw", [dict(reads_per_locus=40, n_str_alleles=6, seed=31),
                                dict(reads_per_locus=30, n_str_alleles=5, n_flank_opts=2, seed=32),
                                dict(reads_per_locus=24, n_str_alleles=8, read_len=100, flank_len=35, str_bp=30, seed=33)])
def test_oracle_trace_matches_compiled_reference(kw):
    ora = capi.load_oracle(); ref = capi.load_ref()
    sb = capi.SynthBatch(n_loci=1, **kw)
    _, seeds = capi.run_align(ora, "oracle_", sb.ptr)
    A = sb.n_out // sb.n_reads
    rng = np.random.default_rng(kw["seed"])
    rr, aa = [], []
    for r in range(sb.n_reads):
        if seeds[r] >= 0:
            for k in rng.choice(A, size=min(A, 3), replace=False):
                rr.append(r); aa.append(int(k))
    h2r = capi.ref_hap_aln_info(ref, sb.ptr, A)
    want = capi.run_trace(ref, "ref_", sb.ptr, rr, aa, cap=1 << 20)
    got = capi.run_trace(ora, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 20)
    util.assert_traces_equal(got, want, str(kw))
