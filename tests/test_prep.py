"""CPU: host logic of the product library (no device): C-ABI surface, seeds, allele order, haplotype rows."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from hipstr_amd import capi
import util
from util import batch_from_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ALIGN = sorted(glob.glob(os.path.join(GOLD, "align_*.npz")))


def test_library_exports_every_declared_symbol(hmm_host):
    """The C-ABI library loads and exports every function include/hipstr_hmm.h declares — the drop-in boundary, which has no hipstr_debug_*
    entry any more (round 6) — and, in the default build, the diagnostics of include/hipstr_hmm_debug.h."""
    hdr = open(os.path.join(ROOT, "include", "hipstr_hmm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(hipstr_[a-z0-9_]+)\s*\(", hdr))
    assert not any(n.startswith("hipstr_debug_") for n in names)
    dbg = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hipstr_hmm_debug.h")).read(), flags=re.S)
    dbg_names = set(re.findall(r"\b(hipstr_debug_[a-z0-9_]+)\s*\(", dbg))
    assert len(dbg_names) >= 12
    for n in sorted(dbg_names):
        assert hasattr(hmm_host, n), "missing diagnostics export " + n
    assert {"hipstr_hmm_init", "hipstr_hmm_upload", "hipstr_hmm_align", "hipstr_hmm_fetch", "hipstr_hmm_process_reads",
            "hipstr_post_run", "hipstr_calc_seed_bases", "hipstr_last_error"} <= names
    for n in sorted(names):
        assert hasattr(hmm_host, n), "missing export " + n


def test_no_cpu_fallback_without_device(hmm_host):
    """Without a GPU the compute entry points fail loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = np.load(os.path.join(GOLD, "align_kat_survey.npz")); b = batch_from_dict(d)
    probs = np.zeros(4); seeds = np.zeros(1, np.int32)
    rc = hmm_host.hipstr_hmm_process_reads(b.ptr, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p))
    assert rc != 0 and b"no HIP device" in hmm_host.hipstr_last_error()
    assert np.all(probs == 0)


@pytest.mark.parametrize("path", ALIGN, ids=[os.path.basename(p)[6:-4] for p in ALIGN])
def test_seed_bases_match_reference(hmm_host, path):
    d = np.load(path); b = batch_from_dict(d)
    n = len(d["expect_seeds"])
    seeds = np.full(max(n, 1), -9, np.int32)
    assert hmm_host.hipstr_calc_seed_bases(b.ptr, seeds.ctypes.data_as(capi._i32p)) == 0
    mask = d["realign_read"].astype(bool) if d["realign_read"].size else np.ones(n, bool)
    assert np.array_equal(seeds[:n][mask], d["expect_seeds"][mask])


def test_bad_cigar_is_an_error(hmm_host):
    from util import simple_locus
    b, _ = simple_locus("ACGTTGCATGCATGACC", ["GA" * 6], "TTGACCGTAGGCTAGG", 2, [("ACGTTGCATGCATGACCGAGA", None, 0, True, [("M", 21)])])
    b.finalize()
    seeds = np.zeros(1, np.int32)
    assert hmm_host.hipstr_calc_seed_bases(b.ptr, seeds.ctypes.data_as(capi._i32p)) != 0


def _one_read_locus(d, l):
    """Locus l of a fixture as its own one-locus batch with its first read only."""
    from hipstr_amd.capi import Batch
    nopts = d["blk_nopts"][3 * l:3 * l + 3]; base = int(d["blk_nopts"][:3 * l].sum())
    seq = d["seq"].tobytes(); oo = d["opt_off"]
    blocks = []; c = base
    for k in range(3):
        opts = [seq[oo[c + o]:oo[c + o + 1]].decode() for o in range(nopts[k])]; c += nopts[k]
        blocks.append((int(d["blk_start"][3 * l + k]), int(d["blk_end"][3 * l + k]), opts))
    r = int(d["read_off"][l])
    assert d["read_off"][l + 1] > r
    b0, b1 = int(d["base_off"][r]), int(d["base_off"][r + 1]); c0, c1 = int(d["cigar_off"][r]), int(d["cigar_off"][r + 1])
    rd = dict(seq=d["bases"].tobytes()[b0:b1].decode(), qual=d["quals"].tobytes()[b0:b1].decode(), start=int(d["read_start"][r]),
              cigar=[(chr(d["cigar_op"][i]), int(d["cigar_len"][i])) for i in range(c0, c1)])
    b = Batch(); A = b.add_locus(blocks, int(d["period"][l]), d["stutter"][6 * l:6 * l + 6], [rd]); b.finalize()
    return b, blocks, A


@pytest.mark.parametrize("name", ["boundary_homopolymers", "homopolymer_two_copy", "synth_multiflank", "tiny_alleles", "masks"])
def test_haplotype_rows_match_oracle(hmm_host, oracle, name):
    """Homopolymer index and base of every flank row, per allele and side — including rows inherited from an
    earlier allele by the reference's alignment reuse and its run-length-table quirk — equal what the oracle's
    literal simulation of the allele loop uses."""
    d = np.load(os.path.join(GOLD, "align_%s.npz" % name))
    b, blocks, A = _one_read_locus(d, 0)
    _check_rows(hmm_host, oracle, b, blocks, A)


def test_rows_next_to_a_two_copy_homopolymer_allele(hmm_host, oracle):
    """Haplotype::homopolymer_length stops its cross-block extension after one neighbour — unless that neighbour is one run whose table entry
    equals its length (Haplotype.cpp:262-270), which with HapBlock's carried counter (2 (n - 1) for a one-run block) is a TWO-base block: the
    two-copy allele of a homopolymer locus.  The leading-flank rows next to it then depend on the flank behind it as well, and they must not
    be shared with an allele that merely starts with a run of three (found by tools/r06_fuzz_fresh.sh, round 6: the rows were cached by
    (flank option, first base, run) alone)."""
    from util import simple_locus
    pre, suf = "TCAGGATCCATGCATTACGATCAG", "CTGATCGTAATGCATGGATCCTGA"
    lfs = [pre + "ACGTTGCAGG", pre + "ACGTTGCATG"]; rfs = ["GGTACCATGC" + suf, "TGTACCATGC" + suf, "GTTACCATGC" + suf]
    strs = ["GGGGGG", "GG", "GGGAGGGG", "GGG", "GGGGCGGGG", "GGTGG", "G" * 9]
    hap = lfs[0] + strs[0] + rfs[0]
    b, A = simple_locus(lfs[0], strs, rfs[0], 1, [(hap[2:-2], None, 2, True)], lf_opts=lfs[1:], rf_opts=rfs[1:])
    b.finalize()
    blocks = [(0, 0, lfs), (0, 0, strs), (0, 0, rfs)]
    assert A == len(lfs) * len(strs) * len(rfs)
    _check_rows(hmm_host, oracle, b, blocks, A)


def _check_rows(hmm_host, oracle, b, blocks, A):
    nopts = np.array([len(blk[2]) for blk in blocks], np.int32)
    for k in range(A):
        hf = np.zeros(2048, np.int32); hr = np.zeros(2048, np.int32)
        assert oracle.oracle_debug_row_h(b.ptr, k, hf.ctypes.data_as(capi._i32p), hr.ctypes.data_as(capi._i32p), 2048) == 0
        o = np.zeros(3, np.int32)
        oracle.oracle_allele_options(nopts.ctypes.data_as(capi._i32p), k, o.ctypes.data_as(capi._i32p))
        s = [blocks[i][2][o[i]] for i in range(3)]
        for side, hh in ((0, hf), (1, hr)):
            lead, mid, trail = (s[0], s[1], s[2]) if side == 0 else (s[2][::-1], s[1][::-1], s[0][::-1])
            for which, seq, row0 in ((0, lead, 0), (1, trail, len(lead) + len(mid))):
                rows = np.zeros(2048, np.uint32)
                n = hmm_host.hipstr_debug_rows(b.ptr, k, side, which, rows.ctypes.data_as(C.POINTER(C.c_uint32)), 2048)
                assert n == len(seq)
                assert bytes((rows[:n] & 0xff).astype(np.uint8)).decode() == seq
                assert np.all(rows[:n] >> 31 == 1)
                assert np.array_equal((rows[1:n] >> 8) & 15, hh[row0 + 1:row0 + n])     # row 0 of a block carries no transition
                u0 = 0 if which == 0 else len(lead) + 1
                assert np.array_equal((rows[:n] >> 12) & 0xfff, np.arange(u0, u0 + n))


def test_tabulated_closed_form_is_exact_below_its_bound(hmm_host, oracle):
    """hs_str_kernel evaluates a simple visiting list as (lp0 + A) + G from a table entry {A, G, Bnd} (prep.cpp simple_table_entry).
    The claim: for |lp0| < Bnd that is bit for bit what fast_log_sum_exp (mathops.cpp:97-106) returns for the values the reference
    pushes — lp0 once per plain offset (+1), ln(U0) + lp0 for the run at the block's end, ln(tail - stop) + lp0 for the equal-likelihood
    rest.  Checked against the oracle's fast_lse_vec on random lists, bounds and lp0 — most of them far below Bnd as in real data, a
    share right below it, where the margin of the error model is thinnest."""
    oracle.oracle_fast_lse_vec.restype = C.c_double
    oracle.oracle_fast_lse_vec.argtypes = [C.POINTER(C.c_double), C.c_int]
    oracle.oracle_int_log.restype = C.c_double
    oracle.oracle_int_log.argtypes = [C.c_int]
    rng = np.random.default_rng(77)
    ent = (C.c_double * 3)()
    checked = near = 0
    for _ in range(4000):
        tail = int(rng.integers(0, 400))
        U0 = int(rng.integers(0, tail + 1)) if rng.random() < 0.8 else 0
        lim = int(rng.integers(0, tail + 1))
        assert hmm_host.hipstr_debug_simple_table(lim, U0, tail, ent) == 0
        A, G, bnd = ent[0], ent[1], ent[2]
        skip = U0 > 0 and lim > 0
        nplain = max(0, lim - U0)
        stop = 0 if lim <= 0 else (U0 if (U0 > 0 and lim <= U0) else lim)
        for rep in range(12):
            if bnd <= 0:
                break
            mag = min(bnd, 1e12) * (1.0 - 2.0 ** -20) * (1.0 if rep < 2 else 0.0) or float(10.0 ** rng.uniform(-1, 5))
            if mag >= bnd:
                continue
            lp0 = -mag if rng.random() < 0.9 else mag
            vals = [lp0] * (1 + nplain)
            if skip:
                vals.append(oracle.oracle_int_log(U0) + lp0)
            if stop < tail:
                vals.append(oracle.oracle_int_log(tail - stop) + lp0)
            arr = (C.c_double * len(vals))(*vals)
            want = oracle.oracle_fast_lse_vec(arr, len(vals))
            got = (lp0 + A) + G
            assert got == want, (lim, U0, tail, lp0, got, want, bnd)
            checked += 1; near += rep < 2
    assert checked > 30000 and near > 5000


def test_host_preparation_does_not_depend_on_thread_count(hmm_host):
    """prepare_batch builds fragments of consecutive loci on several host threads and merges them (prep.cpp); every pool,
    offset and work item must come out exactly as in the single-threaded pass — also with masks and alternative flanks."""
    import ctypes as C
    for kw in (dict(n_loci=37, reads_per_locus=23, n_str_alleles=9, seed=3),
               dict(n_loci=50, reads_per_locus=11, n_str_alleles=6, n_flank_opts=2, seed=4, mask_rate=0.25),
               dict(n_loci=5, reads_per_locus=40, n_str_alleles=12, seed=5)):
        sb = capi.SynthBatch(**kw)
        digests = set()
        for threads in (1, 2, 3, 8, 13):
            sec = C.c_double(); dig = C.c_uint64()
            assert hmm_host.hipstr_debug_prepare(sb.ptr, threads, C.byref(sec), C.byref(dig)) == 0
            digests.add(dig.value)
        assert len(digests) == 1, kw


def test_str_groups_cover_every_read_side_once(hmm_host):
    """The launch plan packs the read sides of a locus into groups whose columns fill one workgroup (hs_str_group_kernel): every seeded
    read appears once per side that has columns, a group stays within the workgroup's lanes, within one locus, and is not empty."""
    import ctypes as C
    for kw in (dict(n_loci=6, reads_per_locus=57, n_str_alleles=9, seed=11),
               dict(n_loci=3, reads_per_locus=200, n_str_alleles=32, read_len=250, flank_len=110, str_bp=60, seed=12),
               dict(n_loci=9, reads_per_locus=30, n_str_alleles=5, read_len=60, flank_len=25, str_bp=20, seed=13, mask_rate=0.2)):
        sb = capi.SynthBatch(**kw)
        b = sb.ptr.contents
        seeds = np.zeros(sb.n_reads, np.int32)
        hmm_host.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
        lens = np.diff(np.ctypeslib.as_array(b.base_off, shape=(sb.n_reads + 1,)))
        read_off_l = np.ctypeslib.as_array(b.read_off, shape=(sb.n_loci + 1,))
        locus_of = np.searchsorted(read_off_l, np.arange(sb.n_reads), side="right") - 1
        cap_g, cap_r = 4 * sb.n_reads + 8, 4 * sb.n_reads + 8
        side = np.zeros(cap_g, np.int32); cols = np.zeros(cap_g, np.int32); roff = np.zeros(cap_g + 1, np.int32); reads = np.zeros(cap_r, np.int32)
        mx = C.c_int32()
        ng = hmm_host.hipstr_debug_str_groups(sb.ptr, side.ctypes.data_as(capi._i32p), cols.ctypes.data_as(capi._i32p), roff.ctypes.data_as(capi._i32p),
                                              cap_g, reads.ctypes.data_as(capi._i32p), cap_r, C.byref(mx))
        assert ng > 0 and mx.value >= 128
        seen = {}
        for g in range(ng):
            rs = reads[roff[g]:roff[g + 1]]
            assert 1 <= len(rs) <= 16 and len(set(locus_of[rs])) == 1
            n = [int(seeds[r]) if side[g] == 0 else int(lens[r] - seeds[r] - 1) for r in rs]
            assert all(x > 0 for x in n) and sum(n) == cols[g] <= mx.value
            for r in rs:
                assert (int(r), int(side[g])) not in seen
                seen[(int(r), int(side[g]))] = g
        realign = np.ctypeslib.as_array(b.realign_read, shape=(sb.n_reads,)) if b.realign_read else np.ones(sb.n_reads, np.uint8)
        for r in range(sb.n_reads):
            if seeds[r] < 0 or not realign[r]:
                continue
            for sd, n in ((0, int(seeds[r])), (1, int(lens[r] - seeds[r] - 1))):
                assert ((r, sd) in seen) == (n > 0), (r, sd, n)


def test_host_preparation_digests_are_pinned():
    """Round 4 rewrote the host preparation for speed (no allocations per locus, closed forms for periodic blocks, memoised table
    entries, recycled storage) and then moved the constants / tables of periodic STR options and the per-allele records to the device
    (expand_kernels.hip).  The rewrite was held to the round-3 code byte for byte before the layout changed (digests of
    hipstr_debug_prepare over thirteen generator shapes: commit "Host preparation: STR options built without allocations ..." and the
    two after it pass tests/golden/prep_digests.json as generated by the round-3 library); the files checked here pin the layout since —
    with the device tables (default) and with HIPSTR_HOST_TABLES=1 — against accidental changes of any pool, offset or work item.
    tests/test_expand_gpu.py holds the device-built tables to the host-built ones."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, env in (("prep_digests.json", {}), ("prep_digests_host_tables.json", {"HIPSTR_HOST_TABLES": "1"})):
        e = dict(os.environ); e.pop("HIPSTR_HOST_TABLES", None); e.update(env)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "prep_digests.py"), "check", os.path.join(root, "tests", "golden", name)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=e)
        assert r.returncode == 0, (name, r.stdout)


def test_locus_costs_price_interrupted_repeats_and_flanks():
    """hipstr_locus_costs (prep.cpp locus_cost): what shard.split_loci and hipstr_multi_submit balance by.  Host-only."""
    import ctypes as C
    from hipstr_amd import shard
    hmm = capi.load_hmm()
    plain = capi.SynthBatch(n_loci=6, reads_per_locus=30, n_str_alleles=8, seed=3)
    a = util.synth_to_batch(plain).arrays
    c = shard.locus_costs(a)
    P = np.diff(a["read_off"]); A = np.diff(a["hap_off"])
    assert c.shape == (6,) and np.all(c > 0)
    # periodic loci of one shape: the estimate is close to reads x alleles x (1.4 x flank / 60 + 1) x length / 150
    assert np.all(c / (P * A) > 1.0) and np.all(c / (P * A) < 6.0)
    os.environ["HIPSTR_SYNTH_INHERIT"] = "2"
    try:
        inter = capi.SynthBatch(n_loci=6, reads_per_locus=30, n_str_alleles=8, seed=3)
        ci = shard.locus_costs(util.synth_to_batch(inter).arrays)
    finally:
        del os.environ["HIPSTR_SYNTH_INHERIT"]
    assert np.all(ci > 1.5 * c)                                   # two inherited interruptions per allele: priced several times a periodic locus
    # the split follows the cost, not the pair count
    both = np.concatenate([c, ci])
    bounds = shard.split_loci(both, 2)
    assert bounds[1] > 6                                          # the cheap half holds more loci


def test_malformed_tables_are_refused_not_crashed_on():
    """tools/fuzz_malformed.py: a valid batch with one or two random corruptions of its tables (an offset negative / huge / decreasing, a count
    zero / negative / huge, a CIGAR run of length 0, a character off, a period or a stutter parameter out of range) goes through the host
    preparation in a child process — every case must come back (accepted, or refused with a message), none may end the process or hang.
    The boundary takes plain pointers (include/hipstr_hmm.h): what it can and does check is that the tables agree with each other
    (prep.cpp validate_tables), before anything indexes with them."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_malformed.py")
    r = subprocess.run([sys.executable, tool, "160", "5"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0 and "crashes/timeouts 0" in r.stdout, r.stdout[-2000:]


def test_capability_limits_are_per_locus_not_table_errors(hmm_host):
    """validate_tables used to fail the WHOLE call for a locus with more than 1024 options of a block — a consistent table of
    an unsupported locus.  Host entry points that only need consistent tables take such a batch now (the seeds of every locus come back),
    hipstr_locus_costs prices it, and tables that contradict each other (hap_off against the product of the option counts) still fail
    the call."""
    b = util.batch_with_an_oversized_middle_locus()
    n_reads, n_out, out_off = capi.batch_dims(b.ptr)
    seeds = np.full(n_reads, -9, np.int32)
    assert hmm_host.hipstr_calc_seed_bases(b.ptr, seeds.ctypes.data_as(capi._i32p)) == 0, hmm_host.hipstr_last_error()
    assert np.all(seeds >= 0)
    bad = capi.Batch.__new__(capi.Batch); bad.__dict__.update(b.__dict__)
    a = dict(b.arrays); a["hap_off"] = a["hap_off"].copy(); a["hap_off"][2:] += 5
    from hipstr_amd import shard
    wrong = shard.batch_from_arrays(a)
    assert hmm_host.hipstr_calc_seed_bases(wrong.ptr, seeds.ctypes.data_as(capi._i32p)) != 0
    assert b"hap_off does not match" in hmm_host.hipstr_last_error()
