"""GPU: hipstr_stream_* — loci submitted one region at a time (the way the reference's caller produces them,
bam_processor.cpp:550-617), batched and pipelined by the library, results handed back in submission order.  Every submission
must come back exactly as hipstr_hmm_process_reads would have answered it alone (bit-identical rows and seeds, untouched
entries untouched), whatever batching the stream chose."""
import numpy as np
import pytest

from hipstr_amd import capi, shard
import util

pytestmark = pytest.mark.gpu
FILL = -4.25


def _pieces(sb, cuts):
    a = util.synth_to_batch(sb).arrays
    return [shard.batch_from_arrays(shard.subset_arrays(a, lo, hi)) for lo, hi in zip(cuts[:-1], cuts[1:])]


@pytest.mark.parametrize("threshold,slots", [(1, 2), (3000, 3), (1 << 40, 2)])
def test_stream_matches_one_shot_calls(hmm, threshold, slots):
    """40 loci with masks, submitted as 1-3 loci per ticket; thresholds from 'every ticket its own batch' to 'one batch, flushed
    by next()'."""
    sb = capi.SynthBatch(n_loci=40, reads_per_locus=25, n_str_alleles=7, n_flank_opts=2, seed=9, mask_rate=0.2)
    rng = np.random.default_rng(1)
    cuts = [0]
    while cuts[-1] < 40:
        cuts.append(min(40, cuts[-1] + int(rng.integers(1, 4))))
    pieces = _pieces(sb, cuts)
    want = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    st = capi.Stream(hmm, slots=slots, batch_alignments=threshold)
    got = []
    tickets = []
    for i, p in enumerate(pieces):          # interleave submission and collection
        tickets.append(st.submit(p.ptr))
        if i % 5 == 4:
            got.append(st.next(fill=FILL))
    while True:
        r = st.next(fill=FILL)
        if r is None:
            break
        got.append(r)
    assert tickets == list(range(len(pieces))) and [g[0] for g in got] == tickets
    for (t, probs, seeds), (wp, ws) in zip(got, want):
        assert np.array_equal(probs, wp) and np.array_equal(seeds, ws), "ticket %d" % t
    s = st.stats()
    assert s["tickets"] == len(pieces) and s["batches"] >= (len(pieces) if threshold == 1 else 1)
    st.close()


def test_stream_against_the_oracle_and_empty_cases(hmm, oracle):
    sb = capi.SynthBatch(n_loci=6, reads_per_locus=30, n_str_alleles=10, seed=21)
    st = capi.Stream(hmm, batch_alignments=500)
    assert st.next() is None                                   # nothing outstanding
    empty = capi.Batch().finalize()
    pieces = _pieces(sb, list(range(7)))
    t0 = st.submit(pieces[0].ptr); te = st.submit(empty.ptr)   # an empty submission is a ticket like any other
    for p in pieces[1:]:
        st.submit(p.ptr)
    res = []
    while True:
        r = st.next()
        if r is None:
            break
        res.append(r)
    assert [r[0] for r in res] == list(range(7)) and res[te][1].size == 0 and t0 == 0
    want, wseeds = capi.run_align(oracle, "oracle_", sb.ptr)
    assert np.array_equal(np.concatenate([r[1] for r in res]), want) and np.array_equal(np.concatenate([r[2] for r in res]), wseeds)
    st.close()


def test_bad_submission_is_refused_alone(hmm):
    """A locus prepare_batch would refuse is turned away at submit; its neighbours are unaffected."""
    good = capi.SynthBatch(n_loci=2, reads_per_locus=10, n_str_alleles=4, seed=2)
    bad, _ = util.simple_locus("ACGTTGCATGCATGACC", ["GA" * 6, ""], "TTGACCGTAGGCTAGG", 2, [])
    bad.finalize()
    st = capi.Stream(hmm)
    st.submit(good.ptr)
    with pytest.raises(RuntimeError, match="empty STR allele"):
        st.submit(bad.ptr)
    st.submit(good.ptr)
    a = st.next(); b = st.next()
    assert a[0] == 0 and b[0] == 1 and np.array_equal(a[1], b[1]) and st.next() is None
    st.close()


def test_large_stream_full_rate(hmm):
    """200 NS-shaped loci through the stream in tickets of 10: identical to the one-shot result of the whole batch."""
    sb = capi.SynthBatch(n_loci=200, reads_per_locus=500, n_str_alleles=32, seed=20260928)
    want, wseeds = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    pieces = _pieces(sb, list(range(0, 201, 10)))
    st = capi.Stream(hmm, slots=3, batch_alignments=1 << 20)
    for p in pieces:
        st.submit(p.ptr)
    out, seeds = [], []
    while True:
        r = st.next()
        if r is None:
            break
        out.append(r[1]); seeds.append(r[2])
    assert np.array_equal(np.concatenate(out), want) and np.array_equal(np.concatenate(seeds), wseeds)
    assert st.stats()["batches"] >= 3
    st.close()


def test_trim_gives_unused_chunks_back_and_keeps_what_is_in_use(hmm):
    """hipstr_hmm_trim: cached chunks without a block in use return to the driver (a closed stream's workspaces), a resident batch keeps its
    chunks and still works, and the library allocates again afterwards."""
    sb = capi.SynthBatch(n_loci=40, reads_per_locus=200, n_str_alleles=16, seed=404)
    want, wseeds = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    dev = hmm.hipstr_hmm_upload(sb.ptr)                      # stays alive across the trim
    assert dev
    st = capi.Stream(hmm, slots=3)
    st.submit(sb.ptr)
    r = st.next()
    assert np.array_equal(r[1], want)
    st.close()
    freed = hmm.hipstr_hmm_trim()
    assert freed >= 0
    assert hmm.hipstr_hmm_align(dev, None) == 0
    p = np.zeros(sb.n_out); s = np.zeros(sb.n_reads, np.int32)
    assert hmm.hipstr_hmm_fetch(dev, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0
    assert np.array_equal(p, want) and np.array_equal(s, wseeds)
    hmm.hipstr_hmm_free(dev)
    freed2 = hmm.hipstr_hmm_trim()                           # now everything this test took can go
    assert freed + freed2 > 0
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)     # and comes back on demand
    assert np.array_equal(got, want) and np.array_equal(gs, wseeds)


def test_multi_device_blocks_in_global_order(hmm):
    """hipstr_multi_*: contiguous blocks of loci dealt to several device streams, results in global submission order.  On a one-GPU
    box both streams sit on device 0 — the dispatch and ordering logic is the same."""
    import ctypes as C
    lib = hmm
    lib.hipstr_multi_open.restype = C.c_void_p; lib.hipstr_multi_open.argtypes = [C.c_int32, capi._i32p, C.c_int64, C.c_void_p]
    lib.hipstr_multi_submit.restype = C.c_int64; lib.hipstr_multi_submit.argtypes = [C.c_void_p, capi._BP]
    lib.hipstr_multi_flush.restype = C.c_int; lib.hipstr_multi_flush.argtypes = [C.c_void_p]
    lib.hipstr_multi_next_size.restype = C.c_int; lib.hipstr_multi_next_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
    lib.hipstr_multi_next.restype = C.c_int; lib.hipstr_multi_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64), capi._f64p, C.c_int64, capi._i32p, C.c_int64]
    lib.hipstr_multi_close.restype = C.c_int; lib.hipstr_multi_close.argtypes = [C.c_void_p]
    sb = capi.SynthBatch(n_loci=30, reads_per_locus=20, n_str_alleles=6, seed=31, mask_rate=0.1)
    pieces = _pieces(sb, list(range(31)))
    want = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    devs = np.zeros(3, np.int32)
    m = lib.hipstr_multi_open(3, devs.ctypes.data_as(capi._i32p), 600, None)          # blocks of ~5 loci, three streams
    assert m, lib.hipstr_last_error()
    for i, p in enumerate(pieces):
        assert lib.hipstr_multi_submit(m, p.ptr) == i
    for i, (wp, ws) in enumerate(want):
        t = C.c_int64(); no = C.c_int64(); nr = C.c_int64()
        assert lib.hipstr_multi_next_size(m, C.byref(t), C.byref(no), C.byref(nr)) == 0 and t.value == i and no.value == wp.size
        probs = np.full(max(no.value, 1), FILL); seeds = np.full(max(nr.value, 1), -7, np.int32)
        assert lib.hipstr_multi_next(m, C.byref(t), probs.ctypes.data_as(capi._f64p), probs.size, seeds.ctypes.data_as(capi._i32p), seeds.size) == 0, lib.hipstr_last_error()
        assert t.value == i and np.array_equal(probs[:no.value], wp) and np.array_equal(seeds[:nr.value], ws)
    assert lib.hipstr_multi_next(m, None, None, 0, None, 0) == 2
    lib.hipstr_multi_close(m)


def test_multi_deals_blocks_by_estimated_work(hmm):
    """hipstr_multi_submit sends a new block to the device that has been dealt the least WORK (hipstr_locus_costs: interrupted repeats cost
    several times a periodic locus' pairs), not round robin by pair counts: periodic loci first, then loci whose every allele inherits two
    interruptions — the expensive blocks spread over the streams, the dealt totals end up close, results in global order and identical."""
    import ctypes as C
    import os
    lib = hmm; _multi_sigs(lib)
    lib.hipstr_multi_dealt.restype = C.c_int; lib.hipstr_multi_dealt.argtypes = [C.c_void_p, capi._f64p, C.c_int32]
    cheap = capi.SynthBatch(n_loci=24, reads_per_locus=20, n_str_alleles=6, seed=41)
    os.environ["HIPSTR_SYNTH_INHERIT"] = "2"
    try:
        dear = capi.SynthBatch(n_loci=12, reads_per_locus=20, n_str_alleles=6, seed=42)
    finally:
        del os.environ["HIPSTR_SYNTH_INHERIT"]
    pc, pd = _pieces(cheap, list(range(25))), _pieces(dear, list(range(13)))
    cc, cd = shard.locus_costs(util.synth_to_batch(cheap).arrays), shard.locus_costs(util.synth_to_batch(dear).arrays)
    assert cd.mean() > 2.0 * cc.mean()
    # cheap, cheap, dear, cheap, cheap, dear ...: round robin over three streams would hand every dear locus to the third
    pieces, costs = [], []
    for k in range(12):
        pieces += [pc[2*k], pc[2*k + 1], pd[k]]; costs += [cc[2*k], cc[2*k + 1], cd[k]]
    costs = np.array(costs)
    want = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    devs = np.zeros(3, np.int32)
    m = lib.hipstr_multi_open(3, devs.ctypes.data_as(capi._i32p), 100, None)          # every locus (120 pairs) its own block
    assert m, lib.hipstr_last_error()
    for i, p in enumerate(pieces):
        assert lib.hipstr_multi_submit(m, p.ptr) == i
    dealt = np.zeros(3)
    assert lib.hipstr_multi_dealt(m, dealt.ctypes.data_as(capi._f64p), 3) == 3
    assert abs(dealt.sum() - costs.sum()) < 1e-6 * costs.sum()
    assert dealt.max() - dealt.min() <= costs.max() * (1 + 1e-9), dealt               # greedy by least work: never further apart than one block
    rr = np.array([costs[k::3].sum() for k in range(3)])                               # what round robin would have dealt
    assert rr.max() - rr.min() > 3 * (dealt.max() - dealt.min())
    for i, (wp, ws) in enumerate(want):
        t = C.c_int64(); no = C.c_int64(); nr = C.c_int64()
        assert lib.hipstr_multi_next_size(m, C.byref(t), C.byref(no), C.byref(nr)) == 0 and t.value == i
        probs = np.full(max(no.value, 1), FILL); seeds = np.full(max(nr.value, 1), -7, np.int32)
        assert lib.hipstr_multi_next(m, C.byref(t), probs.ctypes.data_as(capi._f64p), probs.size, seeds.ctypes.data_as(capi._i32p), seeds.size) == 0, lib.hipstr_last_error()
        assert np.array_equal(probs[:no.value], wp) and np.array_equal(seeds[:nr.value], ws)
    lib.hipstr_multi_close(m)


def test_submit_each_and_collect(hmm):
    """The C-side region loop: every locus of a shard its own submission, a shard's results collected back to back."""
    sb = capi.SynthBatch(n_loci=25, reads_per_locus=18, n_str_alleles=5, seed=41, mask_rate=0.15)
    want, wseeds = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=FILL)
    st = capi.Stream(hmm, batch_alignments=700)
    assert st.submit_each(sb.ptr) == 0 and st.submit_each(sb.ptr) == 25
    for _ in range(2):
        probs = np.full(sb.n_out, FILL); seeds = np.full(sb.n_reads, -7, np.int32)
        assert st.collect(25, probs, seeds) == (sb.n_out, sb.n_reads)
        assert np.array_equal(probs, want) and np.array_equal(seeds, wseeds)
    assert st.next() is None
    st.close()


def test_submit_each_large_call_with_a_bad_locus(hmm):
    """A large hipstr_stream_submit_each call checks its loci on the host threads and appends them run by run; a locus prepare_batch
    would refuse (here: a CIGAR character calc_seed_base does not know, HapAligner.cpp:309) stops the call there — the loci before it
    are in and come back correct, in order; nothing after it was submitted."""
    import util
    from hipstr_amd import shard
    sb = capi.SynthBatch(n_loci=700, reads_per_locus=4, n_str_alleles=3, seed=43)
    want, wseeds = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=FILL)
    a = dict(util.synth_to_batch(sb).arrays)
    bad_locus = 417
    ops = bytearray(a["cigar_op"]); ops[int(a["cigar_off"][int(a["read_off"][bad_locus]) + 1])] = ord("S"); a["cigar_op"] = bytes(ops)
    bad = shard.batch_from_arrays(a)
    n_reads, n_out, out_off = capi.batch_dims(bad.ptr)
    st = capi.Stream(hmm, batch_alignments=2000)
    with pytest.raises(RuntimeError):
        st.submit_each(bad.ptr)
    st.flush()
    ro = a["read_off"]
    probs = np.full(int(out_off[bad_locus]), FILL); seeds = np.full(int(ro[bad_locus]), -7, np.int32)
    assert st.collect(bad_locus, probs, seeds) == (int(out_off[bad_locus]), int(ro[bad_locus]))
    assert np.array_equal(probs, want[:out_off[bad_locus]]) and np.array_equal(seeds, wseeds[:ro[bad_locus]])
    assert st.next() is None
    # the whole shard, valid: one call, every locus its own ticket
    assert st.submit_each(sb.ptr) == bad_locus
    st.flush()
    probs = np.full(sb.n_out, FILL); seeds = np.full(sb.n_reads, -7, np.int32)
    assert st.collect(700, probs, seeds) == (sb.n_out, sb.n_reads)
    assert np.array_equal(probs, want) and np.array_equal(seeds, wseeds)
    st.close()


def _multi_sigs(lib):
    import ctypes as C
    lib.hipstr_multi_open.restype = C.c_void_p; lib.hipstr_multi_open.argtypes = [C.c_int32, capi._i32p, C.c_int64, C.c_void_p]
    lib.hipstr_multi_submit.restype = C.c_int64; lib.hipstr_multi_submit.argtypes = [C.c_void_p, capi._BP]
    lib.hipstr_multi_flush.restype = C.c_int; lib.hipstr_multi_flush.argtypes = [C.c_void_p]
    lib.hipstr_multi_next_size.restype = C.c_int; lib.hipstr_multi_next_size.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
    lib.hipstr_multi_next.restype = C.c_int; lib.hipstr_multi_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64), capi._f64p, C.c_int64, capi._i32p, C.c_int64]
    lib.hipstr_multi_close.restype = C.c_int; lib.hipstr_multi_close.argtypes = [C.c_void_p]


def test_multi_retry_after_too_small_buffer_keeps_the_order(hmm):
    """A hipstr_multi_next that fails because the caller's buffers are too small (return code 3) consumes nothing: neither the
    device stream's ticket nor the entry of the global order.  The retry and everything after it still pair ticket and result."""
    import ctypes as C
    lib = hmm; _multi_sigs(lib)
    sb = capi.SynthBatch(n_loci=12, reads_per_locus=15, n_str_alleles=5, seed=51)
    pieces = _pieces(sb, list(range(13)))
    want = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    devs = np.zeros(2, np.int32)
    m = lib.hipstr_multi_open(2, devs.ctypes.data_as(capi._i32p), 200, None)          # blocks of ~3 loci alternate between two streams
    assert m, lib.hipstr_last_error()
    for i, p in enumerate(pieces):
        assert lib.hipstr_multi_submit(m, p.ptr) == i
    for i, (wp, ws) in enumerate(want):
        t = C.c_int64(); no = C.c_int64(); nr = C.c_int64()
        assert lib.hipstr_multi_next_size(m, C.byref(t), C.byref(no), C.byref(nr)) == 0 and t.value == i
        probs = np.full(max(no.value, 1), FILL); seeds = np.full(max(nr.value, 1), -7, np.int32)
        if i in (1, 4, 5, 9):               # first try with buffers one element short (incl. a ticket that is the last of its block)
            assert lib.hipstr_multi_next(m, C.byref(t), probs.ctypes.data_as(capi._f64p), probs.size - 1, seeds.ctypes.data_as(capi._i32p), seeds.size) == 3
            assert b"too small" in lib.hipstr_last_error()
        assert lib.hipstr_multi_next(m, C.byref(t), probs.ctypes.data_as(capi._f64p), probs.size, seeds.ctypes.data_as(capi._i32p), seeds.size) == 0, lib.hipstr_last_error()
        assert t.value == i and np.array_equal(probs[:no.value], wp) and np.array_equal(seeds[:nr.value], ws), "ticket %d" % i
    assert lib.hipstr_multi_next(m, None, None, 0, None, 0) == 2
    lib.hipstr_multi_close(m)


def test_take_in_any_order_beyond_the_slots(hmm):
    """One thread submits more batches than the stream has slots and takes the LAST ticket first: the batch of a ticket somebody waits
    for goes out even though every slot is held by batches with uncollected earlier tickets (it used to wait forever).  A ticket can
    be taken once; a too-small buffer leaves it takeable."""
    import ctypes as C
    lib = hmm
    lib.hipstr_stream_take.restype = C.c_int; lib.hipstr_stream_take.argtypes = [C.c_void_p, C.c_int64, capi._f64p, C.c_int64, capi._i32p, C.c_int64]
    sb = capi.SynthBatch(n_loci=8, reads_per_locus=12, n_str_alleles=4, seed=61)
    pieces = _pieces(sb, list(range(9)))
    want = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    st = capi.Stream(hmm, slots=2, batch_alignments=1)          # every ticket its own batch, two slots
    for p in pieces:
        st.submit(p.ptr)
    def take(t, short=0):
        wp, ws = want[t]
        probs = np.full(max(wp.size, 1), FILL); seeds = np.full(max(ws.size, 1), -7, np.int32)
        rc = lib.hipstr_stream_take(st.h, t, probs.ctypes.data_as(capi._f64p), probs.size - short, seeds.ctypes.data_as(capi._i32p), seeds.size)
        return rc, probs[:wp.size], seeds[:ws.size]
    for t in (7, 3, 6):
        rc, probs, seeds = take(t)
        assert rc == 0, lib.hipstr_last_error()
        assert np.array_equal(probs, want[t][0]) and np.array_equal(seeds, want[t][1])
    assert take(7)[0] == 1 and b"collected already" in lib.hipstr_last_error()
    assert take(5, short=1)[0] == 3                           # too small: the ticket stays
    rc, probs, seeds = take(5)
    assert rc == 0 and np.array_equal(probs, want[5][0])
    got = []
    while True:                                               # the rest in order
        r = st.next(fill=FILL)
        if r is None:
            break
        got.append(r[0]); assert np.array_equal(r[1], want[r[0]][0]) and np.array_equal(r[2], want[r[0]][1])
    assert got == [0, 1, 2, 4]
    st.close()


def test_multi_on_every_visible_device(hmm):
    """hipstr_multi_* with one stream per physical device (skipped on a one-GPU box): blocks dealt round robin, results in global order,
    each identical to the one-shot call on device 0."""
    import ctypes as C
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one GPU visible")
    lib = hmm; _multi_sigs(lib)
    sb = capi.SynthBatch(n_loci=6 * n_dev, reads_per_locus=20, n_str_alleles=6, seed=71, mask_rate=0.1)
    pieces = _pieces(sb, list(range(6 * n_dev + 1)))
    want = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    devs = np.arange(n_dev, dtype=np.int32)
    m = lib.hipstr_multi_open(n_dev, devs.ctypes.data_as(capi._i32p), 300, None)
    assert m, lib.hipstr_last_error()
    for i, p in enumerate(pieces):
        assert lib.hipstr_multi_submit(m, p.ptr) == i
    for i, (wp, ws) in enumerate(want):
        t = C.c_int64(); no = C.c_int64(); nr = C.c_int64()
        assert lib.hipstr_multi_next_size(m, C.byref(t), C.byref(no), C.byref(nr)) == 0
        probs = np.full(max(no.value, 1), FILL); seeds = np.full(max(nr.value, 1), -7, np.int32)
        assert lib.hipstr_multi_next(m, C.byref(t), probs.ctypes.data_as(capi._f64p), probs.size, seeds.ctypes.data_as(capi._i32p), seeds.size) == 0, lib.hipstr_last_error()
        assert t.value == i and np.array_equal(probs[:no.value], wp) and np.array_equal(seeds[:nr.value], ws), \
            "piece %d from the multi-device stream differs from the one-shot call on device 0" % i
    # (round 6, first contact with a multi-GPU node) every device was dealt work, and the results above — each computed on whichever
    # device its block went to — equal device 0's bit for bit; the devices are named so a bad placement shows in the test log
    lib.hipstr_multi_dealt.restype = C.c_int; lib.hipstr_multi_dealt.argtypes = [C.c_void_p, capi._f64p, C.c_int32]
    dealt = np.zeros(n_dev)
    assert lib.hipstr_multi_dealt(m, dealt.ctypes.data_as(capi._f64p), n_dev) == n_dev and np.all(dealt > 0), dealt
    for dv in range(n_dev):
        pr = torch.cuda.get_device_properties(dv)
        print("device %d: %s pci %04x:%02x:%02x dealt %.3g" % (dv, pr.name, getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0), dealt[dv]))
    lib.hipstr_multi_close(m)
    # the one-shot call on EVERY device against device 0's results: a device that computes differently is named
    for dv in range(1, n_dev):
        assert lib.hipstr_hmm_init(dv) == 0, lib.hipstr_last_error()
        for i in (0, len(pieces) // 2, len(pieces) - 1):
            gp, gsd = capi.run_align(lib, "hipstr_hmm_", pieces[i].ptr, fill=FILL)
            assert np.array_equal(gp, want[i][0]) and np.array_equal(gsd, want[i][1]), "device %d differs from device 0 on piece %d" % (dv, i)
    assert lib.hipstr_hmm_init(0) == 0        # back to device 0 for the tests that follow
