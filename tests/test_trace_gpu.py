"""GPU: hipstr_hmm_trace (Viterbi traceback of one read against one fixed haplotype, HapAligner.cpp:711-722) through the
C-ABI, against the golden vectors of the compiled reference and against the oracle on seeded loci.  Every output is an
integer, a string or the bit-replicated log-likelihood, so the bar is exact equality."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "trace_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[6:-4] for p in FIXTURES])
def test_trace_matches_golden(hmm, path):
    for b, rr, aa, h2r, exp in util.load_trace_fixture(path):
        got = capi.run_trace(hmm, "hipstr_hmm_", b.ptr, rr, aa, h2r, cap=1 << 20)
        util.assert_traces_equal(got, exp, os.path.basename(path))


def _requests(oracle, sb, per_read, seed):
    _, seeds = capi.run_align(oracle, "oracle_", sb.ptr)
    A = sb.n_out // sb.n_reads
    rng = np.random.default_rng(seed)
    rr, aa = [], []
    for r in range(sb.n_reads):
        if seeds[r] >= 0:
            for k in rng.choice(A, size=min(A, per_read), replace=False):
                rr.append(r); aa.append(int(k))
    return rr, aa


@pytest.mark.parametrize("kw", [
    dict(reads_per_locus=50, n_str_alleles=4, seed=1),
    dict(reads_per_locus=40, n_str_alleles=8, n_flank_opts=2, seed=7),
    dict(reads_per_locus=20, n_str_alleles=16, read_len=250, flank_len=110, str_bp=100, seed=5),       # 2-4 columns per lane
    dict(reads_per_locus=60, n_str_alleles=12, read_len=100, flank_len=35, str_bp=30, seed=3),         # reads overhang the flanks
    dict(reads_per_locus=24, n_str_alleles=6, read_len=250, flank_len=160, str_bp=60, seed=9),
])
def test_trace_matches_oracle_on_seeded_loci(hmm, oracle, kw):
    sb = capi.SynthBatch(n_loci=1, **kw)
    rr, aa = _requests(oracle, sb, 3, kw["seed"])
    h2r = util.synthetic_hap_to_ref(oracle, sb.ptr)
    want = capi.run_trace(oracle, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 21)
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 21)
    util.assert_traces_equal(got, want, str(kw))


@pytest.mark.parametrize("kw", [
    dict(reads_per_locus=8, n_str_alleles=3, read_len=640, flank_len=400, str_bp=40, seed=31),        # sides of up to 527 columns: 7-9 per lane
    dict(reads_per_locus=6, n_str_alleles=3, read_len=900, flank_len=520, str_bp=36, seed=32),        # up to 788: 10-13 per lane
    dict(reads_per_locus=8, n_str_alleles=2, read_len=1024, flank_len=600, str_bp=30, seed=33),       # up to 907 of the forward pass' 1024: 11-15 per lane
])
def test_trace_of_long_read_sides(hmm, oracle, kw):
    """Read sides beyond 384 columns (round 4: the traceback takes what the forward pass takes, sides of up to 1024 bases; the fill kernel's
    tables move to dynamic LDS for 8 / 12 / 16 columns per lane) against the oracle, field by field."""
    sb = capi.SynthBatch(n_loci=1, **kw)
    _, seeds = capi.run_align(oracle, "oracle_", sb.ptr)
    lens = np.diff(np.ctypeslib.as_array(sb.ptr.contents.base_off, shape=(sb.n_reads + 1,)))
    longest = max(max(int(s), int(l - s - 1)) for s, l in zip(seeds, lens) if s >= 0)
    assert longest > 384, longest                          # the case is about the new classes
    rr, aa = _requests(oracle, sb, 2, kw["seed"])
    h2r = util.synthetic_hap_to_ref(oracle, sb.ptr)
    want = capi.run_trace(oracle, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 22)
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 22)
    util.assert_traces_equal(got, want, str(kw))


def test_trace_without_reference_strings_skips_the_stitch(hmm, oracle):
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=12, n_str_alleles=4, seed=12)
    rr, aa = _requests(oracle, sb, 2, 12)
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, None)
    want = capi.run_trace(oracle, "oracle_", sb.ptr, rr, aa, None)
    util.assert_traces_equal(got, want)
    assert all(g["cigar"] == "" and g["aln_str"] == "" for g in got)


def test_trace_is_independent_of_the_workspace_chunking(hmm, oracle, monkeypatch):
    """A small HIPSTR_TRACE_WS_MIB forces several launches; the results must not change."""
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=40, n_str_alleles=6, seed=21)
    rr, aa = _requests(oracle, sb, 2, 21)
    h2r = util.synthetic_hap_to_ref(oracle, sb.ptr)
    whole = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r)
    monkeypatch.setenv("HIPSTR_TRACE_WS_MIB", "2")
    pieces = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r)
    util.assert_traces_equal(pieces, whole)


def test_traced_score_equals_forward_score_of_a_single_allele_locus(hmm):
    """With one allele nothing can be reused, so the forward pass and the traceback score the same alignment."""
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=30, n_str_alleles=1, seed=41)
    probs, seeds = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    rr = [r for r in range(sb.n_reads) if seeds[r] >= 0]
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, [0] * len(rr), None)
    assert [g["ll"] for g in got] == [float(probs[r]) for r in rr]


def test_hap_aln_consumes_exactly_the_read(hmm, oracle):
    """Size-independent property: the operation string spends every read base once ('M', 'I', 'S')."""
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=200, n_str_alleles=8, seed=51)
    rr, aa = _requests(oracle, sb, 1, 51)
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, None, cap=1 << 21)
    b = sb.ptr.contents
    base_off = np.ctypeslib.as_array(b.base_off, shape=(sb.n_reads + 1,))
    for g, r in zip(got, rr):
        assert sum(g["hap_aln"].count(c) for c in "MIS") == base_off[r + 1] - base_off[r]
        assert len(g["flank_left"]) + len(g["flank_right"]) + len(g["str_seq"]) + g["hap_aln"].count("S") == base_off[r + 1] - base_off[r]


def test_trace_requests_of_many_loci_in_one_call(hmm, oracle):
    """req_read indexes the whole batch; every locus is checked against the (one-locus) oracle on its own cut."""
    from hipstr_amd import shard
    sb = capi.SynthBatch(n_loci=5, reads_per_locus=30, n_str_alleles=6, n_flank_opts=2, seed=71)
    whole = util.synth_to_batch(sb)
    a = whole.arrays
    _, seeds = capi.run_align(oracle, "oracle_", sb.ptr)
    rng = np.random.default_rng(71)
    rr, aa, want, h2r_all = [], [], [], []
    for l in range(5):
        r0, r1 = int(a["read_off"][l]), int(a["read_off"][l + 1])
        A = int(a["hap_off"][l + 1] - a["hap_off"][l])
        one = shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1))
        h2r = util.synthetic_hap_to_ref(oracle, one.ptr)
        h2r_all += h2r
        lr = [r for r in range(r0, r1) if seeds[r] >= 0]
        la = [int(rng.integers(A)) for _ in lr]
        want += capi.run_trace(oracle, "oracle_", one.ptr, [r - r0 for r in lr], la, h2r, cap=1 << 20)
        rr += lr; aa += la
    order = rng.permutation(len(rr))                 # requests need not be grouped by locus
    got = capi.run_trace(hmm, "hipstr_hmm_", whole.ptr, [rr[i] for i in order], [aa[i] for i in order], h2r_all, cap=1 << 21)
    util.assert_traces_equal(got, [want[i] for i in order])
    # a few requests of a large batch: only the requested reads' bases travel (less than half of the batch's) — the last locus' reads, one
    # of them against two alleles
    few = [i for i in range(len(rr)) if rr[i] >= int(a["read_off"][4])][:7]
    few = few + few[:1]
    got = capi.run_trace(hmm, "hipstr_hmm_", whole.ptr, [rr[i] for i in few], [aa[i] for i in few], h2r_all, cap=1 << 21)
    util.assert_traces_equal(got, [want[i] for i in few])


def test_trace_errors(hmm, oracle):
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=4, n_str_alleles=2, seed=62)
    with pytest.raises(RuntimeError, match="allele outside"):
        capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, [0], [99], None)
    with pytest.raises(RuntimeError, match="read outside"):
        capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, [99], [0], None)
    with pytest.raises(RuntimeError, match="too small"):
        capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, [0, 1, 2, 3], [0, 0, 0, 0], None, cap=64)
    assert capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, [], [], None) == []


def test_random_shapes_match_oracle(hmm, oracle, monkeypatch):
    """Seeded sweep over generator shapes incl. interrupted repeats (non-simple visiting lists in the traced STR row)."""
    rng = np.random.default_rng(99)
    total = 0
    for _ in range(12):
        monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", str(float(rng.choice([0.0, 0.5, 1.0]))))
        kw = dict(n_loci=1, reads_per_locus=int(rng.integers(2, 24)), n_str_alleles=int(rng.integers(1, 20)), read_len=int(rng.integers(24, 251)),
                  flank_len=int(rng.integers(8, 161)), str_bp=int(rng.integers(4, 121)), n_flank_opts=int(rng.integers(1, 4)), seed=int(rng.integers(1, 1 << 30)))
        sb = capi.SynthBatch(**kw)
        rr, aa = _requests(oracle, sb, 2, 5)
        if not rr:
            continue
        h2r = capi.hap_aln_info(oracle, "oracle_", sb.ptr)
        want = capi.run_trace(oracle, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 21)
        got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 21)
        util.assert_traces_equal(got, want, str(kw))
        total += len(rr)
    assert total > 100
