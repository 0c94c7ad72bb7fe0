"""GPU: inputs beyond the shapes the grouped kernels are built for are computed, not refused (HapAligner sizes its matrices by the
read and the haplotype, HapAligner.cpp:573-602): read sides of up to 1024 columns (a 2 x 300 bp MiSeq read seeded two bases from one
end has a side of 297: it takes the workgroup-per-read STR kernel), STR alleles of up to 2047 bp, and a one-shot batch with an invalid
locus in the middle fails that locus only (hipstr_hmm_process_reads_each).  Everything against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from hipstr_amd import capi, shard
import util

pytestmark = pytest.mark.gpu
FILL = -6.5


def test_300bp_reads_seeded_two_bases_from_an_end(hmm, oracle):
    sb = capi.SynthBatch(n_loci=3, reads_per_locus=24, n_str_alleles=12, read_len=300, flank_len=150, str_bp=48, seed=301)
    lens = np.diff(np.ctypeslib.as_array(sb.ptr.contents.base_off, shape=(sb.n_reads + 1,)))
    assert lens.max() == 300
    rng = np.random.default_rng(5)
    seed_in = np.full(sb.n_reads, -2, np.int32)
    pick = rng.random(sb.n_reads)
    seed_in[pick < 0.3] = 2                                  # right side of len - 3 = 297 columns
    hi = (pick >= 0.3) & (pick < 0.6)
    seed_in[hi] = lens[hi] - 3                               # left side of 297 columns
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=FILL, seed_in=seed_in)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=FILL, seed_in=seed_in)
    assert np.array_equal(gs, ws)
    assert (np.maximum(gs, lens - gs - 1)[gs >= 0] > 256).sum() > 10          # sides beyond a group's 256 columns were really there
    assert np.array_equal(got, want)
    # ... and with the seeds calc_seed_base picks
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=FILL)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=FILL)
    assert np.array_equal(gs, ws) and np.array_equal(got, want)


def _long_allele_batch(n_copies, motif="ACAG", extra=(3, -2), n_reads=10, seed=3):
    """One locus whose alleles include a 1500-bp periodic block (and two neighbours), reads drawn from the short reference allele."""
    rng = np.random.default_rng(seed)
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    left, right = rnd(40), rnd(40)
    ref = motif * 12
    alleles = [ref, motif * n_copies] + [motif * (n_copies + e) for e in extra]
    b = capi.Batch()
    reads = []
    for r in range(n_reads):
        hap = left + (ref if r % 2 == 0 else alleles[1]) + right        # odd reads come from the long allele: flank, then repeat to the end
        start = int(rng.integers(0, 20)); L = int(rng.integers(90, 121))
        seq = hap[start:start + L]
        reads.append(dict(seq=seq, qual="F" * len(seq), start=1000 + start, cigar=[("=", len(seq))]))
    blocks = [(1000, 1040, [left]), (1040, 1040 + len(ref), alleles), (1040 + len(ref), 1080 + len(ref), [right])]
    b.add_locus(blocks, len(motif), util.STUTTER, reads)
    return b.finalize()


def test_1500bp_allele(hmm, oracle):
    b = _long_allele_batch(375)                              # 375 x ACAG = 1500 bp, next to 1512 and 1492 bp
    want, ws = capi.run_align(oracle, "oracle_", b.ptr, fill=FILL)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=FILL)
    assert np.array_equal(gs, ws) and (gs >= 0).sum() >= 5
    assert np.array_equal(got, want)


def test_allele_beyond_the_limit_is_refused_with_a_message(hmm):
    b = _long_allele_batch(520)                              # 2080 bp
    with pytest.raises(RuntimeError):
        capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=FILL)
    assert b"2047" in hmm.hipstr_last_error()


def test_invalid_middle_locus_fails_alone(hmm, oracle):
    """Three loci, the middle one with a read whose CIGAR holds a character calc_seed_base does not know (HapAligner.cpp:309: the
    reference dies): hipstr_hmm_process_reads refuses the batch, hipstr_hmm_process_reads_each computes loci 0 and 2 and leaves the
    middle block untouched."""
    sb = capi.SynthBatch(n_loci=3, reads_per_locus=15, n_str_alleles=5, seed=55)
    a = dict(util.synth_to_batch(sb).arrays)
    r_mid = int(a["read_off"][1]) + 3
    ops = bytearray(a["cigar_op"])
    ops[int(a["cigar_off"][r_mid])] = ord("S")
    a["cigar_op"] = bytes(ops)
    bad = shard.batch_from_arrays(a)
    with pytest.raises(RuntimeError):
        capi.run_align(hmm, "hipstr_hmm_", bad.ptr, fill=FILL)
    n_reads, n_out, out_off = capi.batch_dims(bad.ptr)
    probs = np.full(n_out, FILL); seeds = np.full(n_reads, -7, np.int32); status = np.full(3, -1, np.int32)
    hmm.hipstr_hmm_process_reads_each.restype = C.c_int
    hmm.hipstr_hmm_process_reads_each.argtypes = [capi._BP, capi._f64p, capi._i32p, capi._i32p]
    assert hmm.hipstr_hmm_process_reads_each(bad.ptr, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p), status.ctypes.data_as(capi._i32p)) == 0
    assert list(status) == [0, 1, 0] and b"locus 1" in hmm.hipstr_last_error()
    ro = a["read_off"]
    assert np.all(probs[out_off[1]:out_off[2]] == FILL) and np.all(seeds[ro[1]:ro[2]] == -7)
    for l in (0, 2):
        one = shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1))
        wp, ws = capi.run_align(oracle, "oracle_", one.ptr, fill=FILL)
        assert np.array_equal(probs[out_off[l]:out_off[l + 1]], wp) and np.array_equal(seeds[ro[l]:ro[l + 1]], ws)


def test_locus_whose_reads_need_too_much_lds_fails_alone(hmm, oracle):
    """A read of 1950 bases that starts 50 bases in front of the repeat is seeded near its middle (the midpoint of its long repeat-free
    stretch): both sides pass the per-side limit (1024 columns), but the read's tables and the locus' 1200-bp candidate allele do not fit the 160 KiB of
    LDS a workgroup of the per-read STR kernels has.  The check of a single locus (check_locus: hipstr_stream_submit,
    hipstr_hmm_process_reads_each) must turn that locus away ALONE — it used to pass the check and fail the whole shared batch at upload
     —, the loci around it are computed."""
    rng = np.random.default_rng(12)
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    b = capi.Batch()
    for l in range(3):
        left, right = rnd(100 if l == 1 else 40), rnd(1900 if l == 1 else 40)
        ref = "ACAG" * 10
        L = 1950 if l == 1 else 100
        hap = left + ref + right
        reads = []
        for r in range(4):
            start = (len(left) - 50) if l == 1 else int(rng.integers(0, 15))
            seq = hap[start:start + L]
            reads.append(dict(seq=seq, qual="F" * len(seq), start=1000 + start, cigar=[("=", len(seq))]))
        blocks = [(1000, 1000 + len(left), [left]), (1000 + len(left), 1000 + len(left) + len(ref), [ref, "ACAG" * (300 if l == 1 else 11)]),
                  (1000 + len(left) + len(ref), 1000 + len(hap), [right])]
        b.add_locus(blocks, 4, util.STUTTER, reads)
    bb = b.finalize()
    n_reads, n_out, out_off = capi.batch_dims(bb.ptr)
    probs = np.full(n_out, FILL); seeds = np.full(n_reads, -7, np.int32); status = np.full(3, -1, np.int32)
    hmm.hipstr_hmm_process_reads_each.restype = C.c_int
    hmm.hipstr_hmm_process_reads_each.argtypes = [capi._BP, capi._f64p, capi._i32p, capi._i32p]
    assert hmm.hipstr_hmm_process_reads_each(bb.ptr, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p), status.ctypes.data_as(capi._i32p)) == 0
    assert list(status) == [0, 1, 0] and b"LDS" in hmm.hipstr_last_error(), hmm.hipstr_last_error()
    assert np.all(probs[out_off[1]:out_off[2]] == FILL) and np.all(seeds[4:8] == -7)
    assert np.all(np.isfinite(probs[out_off[0]:out_off[1]])) and np.all(probs[out_off[0]:out_off[1]] != FILL) and np.all(seeds[:4] >= 0) and np.all(seeds[8:] >= 0)
    # the same through the stream: the locus is refused at submission, the others go through
    st = capi.Stream(hmm)
    try:
        with pytest.raises(RuntimeError, match="LDS"):
            st.submit_each(bb.ptr)
    finally:
        st.close()


def _lds_partner_loci():
    """Locus 0: reads of 1750 bases (seeded near the middle) with short alleles; locus 1: short reads with a 2000-bp candidate allele.
    Each fits the per-read STR kernels' 160 KiB of LDS alone; the longest read of one with the longest allele of the other does not
    (layout.h hs_str_kernel_lds_bytes: the LDS of a launch is sized by the BATCH's maxima)."""
    rng = np.random.default_rng(21)
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    b = capi.Batch()
    for l in range(3):
        long_reads = (l == 0)
        left, right = rnd(100 if long_reads else 40), rnd(1700 if long_reads else 40)
        ref = "ACAG" * 10
        hap = left + ref + right
        L = 1750 if long_reads else 100
        reads = []
        for r in range(4):
            start = (len(left) - 50) if long_reads else int(rng.integers(0, 15))
            seq = hap[start:start + L]
            reads.append(dict(seq=seq, qual="F" * len(seq), start=1000 + start, cigar=[("=", len(seq))]))
        alts = [ref, "ACAG" * (500 if l == 1 else 11)]
        blocks = [(1000, 1000 + len(left), [left]), (1000 + len(left), 1000 + len(left) + len(ref), alts),
                  (1000 + len(left) + len(ref), 1000 + len(hap), [right])]
        b.add_locus(blocks, 4, util.STUTTER, reads)
    return b.finalize()


def test_loci_that_fit_the_lds_alone_but_not_together_go_into_separate_batches(hmm, oracle):
    """check_locus tests one locus, the upload sizes LDS from the batch-wide longest read and longest allele — a 1.75 kb-read
    locus and a 2 kb-allele locus each pass and used to fail the batch they shared.  The stream and hipstr_hmm_process_reads_each now close
    a batch before a locus that would push the combined figure over the limit; the plain one-shot call still refuses the combination."""
    bb = _lds_partner_loci()
    a = bb.arrays
    n_reads, n_out, out_off = capi.batch_dims(bb.ptr)
    want = []
    for l in range(3):
        one = shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1))
        want.append(capi.run_align(oracle, "oracle_", one.ptr, fill=FILL))
        gp, gs = capi.run_align(hmm, "hipstr_hmm_", one.ptr, fill=FILL)              # every locus is fine alone
        assert np.array_equal(gs, want[l][1]) and np.array_equal(gp, want[l][0])
    with pytest.raises(RuntimeError):
        capi.run_align(hmm, "hipstr_hmm_", bb.ptr, fill=FILL)                        # the caller's own batch of all three: one launch, does not fit
    assert b"LDS" in hmm.hipstr_last_error(), hmm.hipstr_last_error()
    # process_reads_each: split where the figure would overflow
    probs = np.full(n_out, FILL); seeds = np.full(n_reads, -7, np.int32); status = np.full(3, -1, np.int32)
    hmm.hipstr_hmm_process_reads_each.restype = C.c_int
    hmm.hipstr_hmm_process_reads_each.argtypes = [capi._BP, capi._f64p, capi._i32p, capi._i32p]
    assert hmm.hipstr_hmm_process_reads_each(bb.ptr, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p), status.ctypes.data_as(capi._i32p)) == 0, hmm.hipstr_last_error()
    assert list(status) == [0, 0, 0]
    ro = a["read_off"]
    for l in range(3):
        assert np.array_equal(probs[out_off[l]:out_off[l + 1]], want[l][0]) and np.array_equal(seeds[ro[l]:ro[l + 1]], want[l][1])
    # the stream, every locus a submission: they share batches only where they fit together
    for each in (True, False):
        st = capi.Stream(hmm)
        try:
            if each:
                st.submit_each(bb.ptr)
            else:
                for l in range(3):
                    st.submit(shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1)).ptr)
            st.flush()
            for l in range(3):
                t, gp, gs = st.next(fill=FILL)
                assert t == l and np.array_equal(gp, want[l][0]) and np.array_equal(gs, want[l][1])
            assert st.stats()["batches"] >= 2
        finally:
            st.close()
    # one SUBMISSION is one batch: all three loci as one submission is refused with a message that says what to do
    st = capi.Stream(hmm)
    try:
        with pytest.raises(RuntimeError, match="submit them separately"):
            st.submit(bb.ptr)
    finally:
        st.close()


def test_oversized_middle_locus_fails_alone(hmm, oracle):
    """a locus with 1025 options of a block (the library takes 1024) in the middle of a hipstr_hmm_process_reads_each batch
    and of a hipstr_stream_submit_each call: that locus is refused alone with a message; process_reads_each computes the loci before AND
    behind it, the stream keeps its contract (the loci before the refused one are in).  The table check used to fail the whole call."""
    b = util.batch_with_an_oversized_middle_locus()
    a = b.arrays
    n_reads, n_out, out_off = capi.batch_dims(b.ptr)
    probs = np.full(n_out, FILL); seeds = np.full(n_reads, -7, np.int32); status = np.full(3, -1, np.int32)
    hmm.hipstr_hmm_process_reads_each.restype = C.c_int
    hmm.hipstr_hmm_process_reads_each.argtypes = [capi._BP, capi._f64p, capi._i32p, capi._i32p]
    assert hmm.hipstr_hmm_process_reads_each(b.ptr, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p), status.ctypes.data_as(capi._i32p)) == 0, hmm.hipstr_last_error()
    assert list(status) == [0, 1, 0] and b"locus 1" in hmm.hipstr_last_error() and b"1024" in hmm.hipstr_last_error()
    ro = a["read_off"]
    assert np.all(probs[out_off[1]:out_off[2]] == FILL) and np.all(seeds[ro[1]:ro[2]] == -7)
    want = {}
    for l in (0, 2):
        one = shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1))
        wp, ws = capi.run_align(oracle, "oracle_", one.ptr, fill=FILL)
        assert np.array_equal(probs[out_off[l]:out_off[l + 1]], wp) and np.array_equal(seeds[ro[l]:ro[l + 1]], ws)
        want[l] = (wp, ws)
    with pytest.raises(RuntimeError):                          # the all-or-nothing call refuses the batch, with the locus named
        capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=FILL)
    st = capi.Stream(hmm)                                      # the stream's contract: stops at the refused locus, the loci before it are in
    with pytest.raises(RuntimeError, match="1024"):
        st.submit_each(b.ptr)
    st.submit(shard.batch_from_arrays(shard.subset_arrays(a, 2, 3)).ptr)
    st.flush()
    got = [st.next(fill=FILL) for _ in range(2)]
    assert st.next() is None
    st.close()
    for (t, p, s), l in zip(got, (0, 2)):
        assert np.array_equal(p, want[l][0]) and np.array_equal(s, want[l][1])
