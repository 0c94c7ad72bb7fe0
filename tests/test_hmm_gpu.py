"""GPU: the HIP forward-HMM path (through the C-ABI) against the golden fixtures of the compiled reference,
against the oracle on fresh seeded inputs, and — at BASELINE sizes — through size-independent properties.

Tolerance: the path computes in IEEE double with the reference's operation order and bit-replicates its float
log-sum-exp approximations, so the stated tolerance is |dLL| <= 1e-9 * max(1, |LL|) and the
observed difference is exactly 0; the tests assert the former and report the latter."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
from util import batch_from_dict, simple_locus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALIGN = sorted(glob.glob(os.path.join(GOLD, "align_*.npz")))


def _close(got, want):
    return np.all(np.abs(got - want) <= 1e-9 * np.maximum(1.0, np.abs(want)))


@pytest.mark.parametrize("path", ALIGN, ids=[os.path.basename(p)[6:-4] for p in ALIGN])
def test_golden_fixtures(hmm, path):
    d = np.load(path); b = batch_from_dict(d)
    sent = float(d["sentinel"][0])
    probs, seeds = capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=sent)
    want = d["expect_aln_probs"]
    assert np.array_equal(seeds[d["expect_seeds"] != -7], d["expect_seeds"][d["expect_seeds"] != -7])
    assert np.array_equal(probs == sent, want == sent), "untouched-entry contract (HapAligner.cpp:326-329, 615-619)"
    assert _close(probs, want)
    assert np.array_equal(probs, want), "expected bit-exact agreement; max|diff| = %g" % np.max(np.abs(probs - want))


@pytest.mark.parametrize("kw", [
    dict(n_loci=1, reads_per_locus=50, n_str_alleles=4, seed=1),                                     # BASELINE configs[0]
    dict(n_loci=24, reads_per_locus=40, n_str_alleles=32, seed=2),                                   # configs[1] shape, shrunk
    dict(n_loci=6, reads_per_locus=24, n_str_alleles=6, n_flank_opts=3, seed=3, mask_rate=0.3),
    dict(n_loci=3, reads_per_locus=12, n_str_alleles=48, read_len=250, flank_len=110, str_bp=100, seed=4),   # configs[4] shape, shrunk
    dict(n_loci=10, reads_per_locus=30, n_str_alleles=10, read_len=101, flank_len=35, str_bp=28, seed=5),
    dict(n_loci=2, reads_per_locus=600, n_str_alleles=8, seed=6),                                    # configs[2] shape: many reads per locus
])
def test_matches_oracle_on_seeded_loci(hmm, oracle, kw):
    sb = capi.SynthBatch(**kw)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
    assert np.array_equal(gs, ws)
    assert _close(got, want)
    assert np.array_equal(got, want), "max|diff| = %g" % np.max(np.abs(got - want))


def test_resident_batch_api(hmm, oracle):
    """upload / align / fetch with the batch resident in HBM; repeated passes are idempotent."""
    sb = capi.SynthBatch(n_loci=5, reads_per_locus=20, n_str_alleles=8, seed=9)
    dev = hmm.hipstr_hmm_upload(sb.ptr)
    assert dev, hmm.hipstr_last_error()
    outs = []
    for _ in range(2):
        assert hmm.hipstr_hmm_align(dev, None) == 0
        p = np.zeros(sb.n_out); s = np.zeros(sb.n_reads, np.int32)
        assert hmm.hipstr_hmm_fetch(dev, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0
        outs.append(p)
    ms = capi.C.c_float(0)
    assert hmm.hipstr_hmm_align_timed(dev, 3, capi.C.byref(ms), None) == 0 and ms.value > 0
    hmm.hipstr_hmm_free(dev)
    want, _ = capi.run_align(oracle, "oracle_", sb.ptr)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], want)


def test_error_paths(hmm):
    b, _ = simple_locus("ACGTTGCATGCATGACC", ["GA" * 6, ""], "TTGACCGTAGGCTAGG", 2, [])
    b.finalize()
    p = np.zeros(1); s = np.zeros(1, np.int32)
    assert hmm.hipstr_hmm_process_reads(b.ptr, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) != 0
    assert b"empty STR allele" in hmm.hipstr_last_error()
    b2, _ = simple_locus("ACGTTGCATGCATGACC", ["GA" * 6], "TTGACCGTAGGCTAGG", 2, [("ACGTTGCATGCATGACCGAGA", None, 0, True, [("S", 21)])])
    b2.finalize()
    assert hmm.hipstr_hmm_process_reads(b2.ptr, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) != 0


def test_empty_batch(hmm):
    b = capi.Batch().finalize()
    p = np.zeros(1); s = np.zeros(1, np.int32)
    assert hmm.hipstr_hmm_process_reads(b.ptr, p.ctypes.data_as(capi._f64p), s.ctypes.data_as(capi._i32p)) == 0


def test_full_size_properties(hmm, oracle):
    """BASELINE configs[1] scale slice (64 loci x 500 reads x 32 alleles = 1.0M alignments) checked through
    properties that do not need the oracle at full size:
      * a read's row does not depend on the rest of the batch: the first loci re-run alone give identical rows;
      * duplicate (pooled-identical) reads get identical rows;
      * every log-likelihood is finite and <= 0 (compute_aln_logprob's assert, HapAligner.cpp:229);
      * a strided sample of reads agrees with the oracle bit for bit."""
    big = capi.SynthBatch(n_loci=64, reads_per_locus=500, n_str_alleles=32, seed=20260928)
    got, seeds = capi.run_align(hmm, "hipstr_hmm_", big.ptr)
    assert np.all(np.isfinite(got)) and np.all(got <= 1e-10)
    small = capi.SynthBatch(n_loci=3, reads_per_locus=500, n_str_alleles=32, seed=20260928)
    g2, s2 = capi.run_align(hmm, "hipstr_hmm_", small.ptr)
    assert np.array_equal(g2, got[:small.n_out]) and np.array_equal(s2, seeds[:small.n_reads])
    one = capi.SynthBatch(n_loci=1, reads_per_locus=500, n_str_alleles=32, seed=20260928)   # keep alive: .ptr borrows from it
    w2, ws2 = capi.run_align(oracle, "oracle_", one.ptr)
    assert np.array_equal(w2, got[:w2.size])
    # duplicates: identical read bytes + CIGAR within a locus -> identical rows
    from util import synth_to_batch
    d = synth_to_batch(small).arrays
    A = int(d["hap_off"][1]); rows = got[:500 * A].reshape(500, A)
    seen = {}
    for r in range(500):
        key = (d["bases"][d["base_off"][r]:d["base_off"][r + 1]], d["quals"][d["base_off"][r]:d["base_off"][r + 1]], int(seeds[r]))
        if key in seen:
            assert np.array_equal(rows[r], rows[seen[key]])
        seen[key] = r


def test_north_star_batch_full_size(hmm, oracle):
    """The whole BASELINE configs[1] / north-star batch — 1000 loci x 500 reads x 32 alleles, 15.9 M alignments in one call:
      * every log-likelihood finite and <= 0 (compute_aln_logprob's assert, HapAligner.cpp:229), every seed as the host computes it;
      * rows do not depend on batch composition: ten loci spread over the batch, each re-run ALONE, give identical rows;
      * a strided 1 % sample — locus 7, 107, 207, ... regenerated alone from (seed, index) — agrees with the oracle bit for bit;
      * running the resident batch a second time reproduces the first result bit for bit."""
    NL, P, A = 1000, 500, 32
    big = capi.SynthBatch(n_loci=NL, reads_per_locus=P, n_str_alleles=A, seed=20260928)
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
    for l in range(7, NL, 100):
        one = capi.SynthBatch(n_loci=1, reads_per_locus=P, n_str_alleles=A, seed=20260928, first_locus=l)
        lo, hi = int(big.out_off[l]), int(big.out_off[l + 1])
        alone, s1 = capi.run_align(hmm, "hipstr_hmm_", one.ptr)
        assert np.array_equal(alone, got[lo:hi]) and np.array_equal(s1, seeds[l * P:(l + 1) * P]), "locus %d depends on its batch" % l
        want, ws = capi.run_align(oracle, "oracle_", one.ptr)
        assert np.array_equal(want, got[lo:hi]) and np.array_equal(ws, s1), "locus %d differs from the oracle" % l


def test_config5_stress_batch_full_size(hmm, oracle):
    """The whole BASELINE configs[4] stress batch as bench.py --workload c5 runs it — 256 loci x 200 reads of 250 bp x 128 alleles,
    ~100-bp blocks, 110-bp flanks (two rounds of four 15-row bands per sweep), 5.8 M alignments in one call: all finite and <= 0, a
    second run identical, three loci spread over the batch identical when re-run ALONE, and two of them bit-equal to the oracle."""
    NL, P, A, kw = 256, 200, 128, dict(read_len=250, flank_len=110, str_bp=100)
    big = capi.SynthBatch(n_loci=NL, reads_per_locus=P, n_str_alleles=A, seed=20260928, **kw)
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
    for n, l in enumerate((5, 130, 251)):
        one = capi.SynthBatch(n_loci=1, reads_per_locus=P, n_str_alleles=A, seed=20260928, first_locus=l, **kw)
        lo, hi = int(big.out_off[l]), int(big.out_off[l + 1])
        alone, s1 = capi.run_align(hmm, "hipstr_hmm_", one.ptr)
        assert np.array_equal(alone, got[lo:hi]) and np.array_equal(s1, seeds[l * P:(l + 1) * P]), "locus %d depends on its batch" % l
        if n < 2:
            want, ws = capi.run_align(oracle, "oracle_", one.ptr)
            assert np.array_equal(want, got[lo:hi]) and np.array_equal(ws, s1), "locus %d differs from the oracle" % l


def test_dropin_adapter_against_reference_objects(hmm):
    """integration/HapAlignerMI355X — HapAligner's interface on the reference's own Haplotype/Alignment objects — next to
    the reference's CPU HapAligner in one process (oracle/_ref/dropin_check, prebuilt where the HipSTR tree is mounted)."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "dropin_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_check not built (needs the reference tree at build time)")
    for args in (["6", "24", "8", "2"], ["3", "40", "32", "1"]):
        out = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
        assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout


def test_workspace_chunking(hmm, oracle, monkeypatch):
    """The phase kernels hand results over through HBM workspaces; a batch larger than the workspace budget is cut into
    chunks of reads.  Force many tiny chunks and check the result is unchanged."""
    sb = capi.SynthBatch(n_loci=6, reads_per_locus=40, n_str_alleles=12, n_flank_opts=2, seed=44, mask_rate=0.15)
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-9.5)
    monkeypatch.setenv("HIPSTR_WS_GIB", "0.0005")        # ~0.5 MiB per workspace -> a few reads per chunk
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-9.5)
    monkeypatch.delenv("HIPSTR_WS_GIB")
    assert np.array_equal(gs, ws) and np.array_equal(got, want)


def test_imperfect_repeats_and_long_periods(hmm, oracle):
    """Interrupted repeats (generic visiting lists incl. lists longer than one 64-entry bundle), period 9, alleles near the
    256 bp limit: the slow paths of the STR kernel."""
    import random
    rnd = random.Random(5)
    lf = "".join(rnd.choice("ACGT") for _ in range(48)); rf = "".join(rnd.choice("ACGT") for _ in range(48))
    motif = "ACGGTTCAG"
    strs = [motif * 8, motif * 9, motif * 7 + "ACGGTTCAT", "ACGATTCAG" + motif * 7, motif * 3 + "T" + motif * 4]
    long_imp = "".join(rnd.choice("ACGT") for _ in range(230))                  # aperiodic 230 bp block: lists of ~230 entries
    strs.append(long_imp)
    hap = lf + strs[0] + rf
    reads = [(hap[s:s + 150], None, s, True) for s in (0, 3, 9, 14, 18)]
    reads.append((lf[10:] + long_imp[:120], None, 10, True, [("=", 38), ("X", 120)]))
    b, A = simple_locus(lf, strs, rf, 9, reads)
    b.finalize()
    got, gs = capi.run_align(hmm, "hipstr_hmm_", b.ptr)
    want, ws = capi.run_align(oracle, "oracle_", b.ptr)
    assert np.array_equal(gs, ws) and np.array_equal(got, want), np.max(np.abs(got - want))


def test_random_shapes_match_oracle(hmm, oracle, monkeypatch):
    """A seeded sweep over generator shapes (read / flank / STR sizes, allele counts, flank options, masks, interrupted repeats): the
    unclamped pointer-stepped loops of the STR kernel rely on masking, so odd sizes are where a slip would show."""
    rng = np.random.default_rng(2026)
    total = 0
    for _ in range(16):
        monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", str(float(rng.choice([0.0, 0.05, 0.3, 1.0]))))
        kw = dict(n_loci=int(rng.integers(1, 4)), reads_per_locus=int(rng.integers(1, 30)), n_str_alleles=int(rng.integers(1, 33)),
                  read_len=int(rng.integers(24, 251)), flank_len=int(rng.integers(8, 161)), str_bp=int(rng.integers(4, 121)),
                  n_flank_opts=int(rng.integers(1, 4)), seed=int(rng.integers(1, 1 << 30)), mask_rate=float(rng.choice([0.0, 0.3])))
        sb = capi.SynthBatch(**kw)
        want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
        assert np.array_equal(gs, ws) and np.array_equal(got, want), kw
        total += got.size
    assert total > 10000


@pytest.mark.parametrize("env", [("HIPSTR_DEBUG_REDO", "1"), ("HIPSTR_DEBUG_REDO", "3"), ("HIPSTR_DEBUG_BND_SCALE", "1e-6"),
                                 ("HIPSTR_DEBUG_BND_SCALE", "0")], ids=["redo_all", "redo_third", "bound_1e-6", "bound_0"])
def test_tabulated_closed_form_and_its_redo_path(hmm, oracle, monkeypatch, env):
    """hs_str_kernel evaluates simple visiting lists from a table that is exact only while |lp0| stays below a per-entry bound; chunks of
    columns that cannot promise that are marked and re-done the long way by hs_str_kernel_generic.  Force that path (every chunk, every
    third chunk, a bound shrunk by 1e6, a bound of zero) on perfect, interrupted and masked inputs: the result must not move."""
    monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", "0.1")
    cases = [dict(n_loci=5, reads_per_locus=30, n_str_alleles=32, seed=71),
             dict(n_loci=3, reads_per_locus=20, n_str_alleles=9, n_flank_opts=2, seed=72, mask_rate=0.3),
             dict(n_loci=2, reads_per_locus=12, n_str_alleles=128, read_len=250, flank_len=120, str_bp=90, seed=73)]
    for kw in cases:
        sb = capi.SynthBatch(**kw)
        want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
        monkeypatch.setenv(*env)
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
        monkeypatch.delenv(env[0])
        assert np.array_equal(gs, ws) and np.array_equal(got, want), (env, kw)


@pytest.mark.parametrize("kw,env", [
    (dict(n_loci=3, reads_per_locus=40, n_str_alleles=12, read_len=257, flank_len=120, str_bp=40, seed=81), None),     # sides up to the supported 256 columns: one read fills a group
    (dict(n_loci=4, reads_per_locus=90, n_str_alleles=20, read_len=60, flank_len=28, str_bp=24, seed=82), None),       # short sides: many reads per group, some under six periods
    (dict(n_loci=3, reads_per_locus=64, n_str_alleles=32, read_len=200, flank_len=90, str_bp=48, seed=83), None),      # one long and one or two short sides per group
    (dict(n_loci=6, reads_per_locus=50, n_str_alleles=32, seed=84), ("HIPSTR_STR_GROUP", "0")),                       # one workgroup per read for every side
], ids=["longest_sides", "short_sides", "mixed_sides", "per_read_kernel"])
def test_str_groups_and_their_fallback(hmm, oracle, monkeypatch, kw, env):
    """hs_str_group_kernel packs the columns of several reads of a locus side into one workgroup; a side with more columns than a
    group holds would stay with hs_str_kernel (one workgroup per read; the library accepts sides up to 256 columns, exactly a group,
    so that path is reached only through HIPSTR_STR_GROUP=0, which selects it for every side)."""
    monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", "0.05")
    sb = capi.SynthBatch(**kw)
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
    if env:
        monkeypatch.setenv(*env)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    assert np.array_equal(gs, ws) and np.array_equal(got, want), kw


@pytest.mark.parametrize("kw,env", [
    (dict(n_loci=4, reads_per_locus=60, n_str_alleles=32, seed=91), None),                                             # NS shape: three reads per group
    (dict(n_loci=3, reads_per_locus=30, n_str_alleles=24, read_len=100, flank_len=30, str_bp=90, seed=92), None),      # blocks longer than the read sides
    (dict(n_loci=2, reads_per_locus=24, n_str_alleles=16, read_len=300, flank_len=140, str_bp=50, seed=93), None),     # sides beyond a group's 256 columns: these reads stay with hs_str_kernel_generic
    (dict(n_loci=3, reads_per_locus=40, n_str_alleles=20, n_flank_opts=3, mask_rate=0.3, seed=94), None),              # several lead slots, masked alleles and reads
    (dict(n_loci=3, reads_per_locus=40, n_str_alleles=32, seed=95), ("HIPSTR_DEBUG_REDO", "2")),                      # half of the wavefront-alleles re-done by the generic kernel
    (dict(n_loci=3, reads_per_locus=40, n_str_alleles=32, str_bp=14, seed=96), None),                                  # short blocks: fewer deletion sizes than six
], ids=["ns", "long_blocks", "long_sides", "flank_options_masks", "redo_half", "short_blocks"])
def test_interrupted_repeats_in_the_grouped_kernel(hmm, oracle, monkeypatch, kw, env):
    """hs_str_group_kernel_pw: every alternative allele carries a random substitution (HIPSTR_SYNTH_IMPERFECT=1.0), so its visiting lists are
    piecewise simple — one or two breaks — and are evaluated by pw_eval_grp from scalar descriptor slots next to the tabulated lists of the
    same allele.  Against the oracle, bit for bit."""
    monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", "1.0")
    sb = capi.SynthBatch(**kw)
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
    if env:
        monkeypatch.setenv(*env)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    assert np.array_equal(gs, ws) and np.array_equal(got, want), kw


@pytest.mark.parametrize("lf_len,rf_len", [(1, 40), (40, 1), (2, 33), (20, 21), (21, 20), (41, 61)])
def test_flank_heights_around_the_band_size(hmm, oracle, lf_len, rf_len):
    """The banded sweeps of hs_lead_kernel / hs_trail_kernel cut a flank into bands of <= 20 rows; a one-base flank is a block with
    matrix row 0 (or the "must be followed by a match" row) and nothing else.  Flank lengths of 1, 2, exactly one band, one band + 1, two
    bands + 1 and three bands + 1, on both sides, reads of several lengths sharing a wavefront."""
    import random
    rnd = random.Random(1000 * lf_len + rf_len)
    lf = "".join(rnd.choice("ACGT") for _ in range(lf_len)); rf = "".join(rnd.choice("ACGT") for _ in range(rf_len))
    strs = ["ACG" * k for k in (6, 5, 7, 9)] + ["ACG" * 3 + "ATG" + "ACG" * 3]
    hap = lf + strs[0] + rf
    reads = []
    for s in range(0, max(1, len(hap) - 24), 3):
        for ln in (24, 31, len(hap) - s):
            if s + ln <= len(hap):
                reads.append((hap[s:s + ln], None, s, True))
    b, A = simple_locus(lf, strs, rf, 3, reads)
    b.finalize()
    want, ws = capi.run_align(oracle, "oracle_", b.ptr, fill=-2.5)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=-2.5)
    assert (ws >= 0).sum() > 0, "no read of this shape has a seed: the case tests nothing"
    assert np.array_equal(gs, ws) and np.array_equal(got, want), np.max(np.abs(got - want))


@pytest.mark.parametrize("n_alleles", [3, 9, 20, 40])
def test_flank_rows_of_any_character(hmm, oracle, n_alleles):
    """A flank row is compared with a read base as a character (HapAligner.cpp:144-153): N and lower case in the flanks and N in the reads
    take the same path as A, C, G, T (the compiled reference agrees with the oracle on these inputs).  The trailing-flank kernel writes a column's emissions to an LDS table by band row (groups of
    8 and more alleles; the lanes of a read compare the read base with their row's base) or selects per cell (smaller groups): both here."""
    import random
    rnd = random.Random(77 + n_alleles)
    def flank(n):
        return "".join(rnd.choice("ACGT" if rnd.random() < 0.8 else "Nacgtn") for _ in range(n))
    lf, rf = flank(37), flank(44)
    strs = ["ACAG" * k for k in range(5, 5 + n_alleles)]
    hap = lf + strs[0] + rf
    reads = []
    for s in range(0, len(hap) - 30, 2):
        for ln in (30, 45, min(70, len(hap) - s)):
            if s + ln <= len(hap):
                seq = list(hap[s:s + ln].upper())
                for k in range(len(seq)):
                    if rnd.random() < 0.04: seq[k] = rnd.choice("ACGTN")
                reads.append(("".join(seq), "".join(chr(33 + rnd.randint(2, 40)) for _ in seq), s, True))
    b, A = simple_locus(lf, strs, rf, 4, reads)
    b.finalize()
    want, ws = capi.run_align(oracle, "oracle_", b.ptr, fill=-2.5)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=-2.5)
    assert (ws >= 0).sum() > 20, "too few reads of this shape have a seed: the case tests nothing"
    assert np.array_equal(gs, ws) and np.array_equal(got, want), np.max(np.abs(got - want))


@pytest.mark.parametrize("mask", [[0, 0, 0], [1, 0, 0], [0, 0, 1], [0, 1, 1]])
def test_allele_masks_down_to_none(hmm, oracle, mask):
    """realign_to_haplotype masks (HapAligner.cpp:615-619) including the empty one: the kernels of every phase must cope with a locus that
    has no allele to align (no STR work, no trailing-flank groups), and masked entries keep the caller's values."""
    lf = "ACGTTGCATGCATGACCTTGACGGT"; rf = "TTGACCGTAGGCTAGGCATTACGGA"
    strs = ["GA" * 6, "GA" * 7, "GA" * 5]
    hap = lf + strs[0] + rf
    reads = [(hap[s:s + 40], None, s, True) for s in (0, 3, 8, 15)]
    b, A = simple_locus(lf, strs, rf, 2, reads, realign_hap=mask)
    b.finalize()
    want, ws = capi.run_align(oracle, "oracle_", b.ptr, fill=-7.5)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=-7.5)
    assert np.array_equal(gs, ws) and np.array_equal(got, want)
    assert np.all((got == -7.5).reshape(-1, 3)[:, [i for i, m in enumerate(mask) if not m]])


@pytest.mark.parametrize("inherit,imperfect,kw", [
    (2, "0.0", dict(n_loci=4, reads_per_locus=60, n_str_alleles=32, seed=191)),                                        # NS shape, two interruptions in every allele: 3-4 breaks per list
    (3, "0.0", dict(n_loci=4, reads_per_locus=60, n_str_alleles=32, seed=192)),                                        # three: up to six breaks
    (3, "1.0", dict(n_loci=3, reads_per_locus=40, n_str_alleles=24, seed=193)),                                        # + a random substitution per alt allele: lists beyond six breaks are replayed
    (2, "0.3", dict(n_loci=3, reads_per_locus=30, n_str_alleles=24, read_len=100, flank_len=30, str_bp=90, seed=194)), # blocks longer than the read sides
    (3, "0.0", dict(n_loci=3, reads_per_locus=40, n_str_alleles=20, n_flank_opts=3, mask_rate=0.3, seed=195)),         # lead slots, masks
    (2, "0.0", dict(n_loci=2, reads_per_locus=24, n_str_alleles=16, read_len=300, flank_len=140, str_bp=60, seed=196)), # sides beyond a group: the per-read kernel replays these lists
], ids=["inherit2", "inherit3", "inherit3_plus_random", "long_blocks", "flank_options_masks", "long_sides"])
def test_inherited_interruptions_k_level_closed_form(hmm, oracle, monkeypatch, inherit, imperfect, kw):
    """hs_str_group_kernel_rp: the reference allele carries two or three interrupted repeat units and every candidate inherits them
    (HIPSTR_SYNTH_INHERIT), so the visiting lists have three to six breaks and are evaluated by the K-level piecewise closed form
    (pwk_eval_grp, layout.h HS_SHAPE_PWK) — lists with still more breaks by the entry-by-entry replay in the same kernel.  Against the oracle,
    bit for bit; the kind-3 share of the batch is checked so that the case keeps exercising the kernel."""
    import ctypes as C
    monkeypatch.setenv("HIPSTR_SYNTH_INHERIT", str(inherit))
    monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", imperfect)
    sb = capi.SynthBatch(**kw)
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    assert np.array_equal(gs, ws) and np.array_equal(got, want), kw
    dev = hmm.hipstr_hmm_upload(sb.ptr)
    assert dev
    kinds = (C.c_int64 * 4)()
    hmm.hipstr_debug_allele_kinds(dev, kinds)
    hmm.hipstr_hmm_free(dev)
    if kw.get("read_len", 150) <= 257:
        assert kinds[3] > 0.3 * sum(kinds), list(kinds)


@pytest.mark.parametrize("mode", ["spin", "sleep", "yield"])
def test_wait_modes_of_the_one_shot_calls(mode):
    """HIPSTR_WAIT (api_internal.h wait_stream): the one-shot entry points poll their stream and yield the core (default), spin in the driver
    (spin) or sleep between polls (sleep); read once per process, so each mode runs in a process of its own, against the oracle."""
    import subprocess, sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from hipstr_amd import capi\n"
            "hmm = capi.load_hmm(); ora = capi.load_oracle(); assert hmm.hipstr_hmm_init(0) == 0\n"
            "sb = capi.SynthBatch(n_loci=3, reads_per_locus=30, n_str_alleles=8, seed=77)\n"
            "want, ws = capi.run_align(ora, 'oracle_', sb.ptr, fill=-3.25)\n"
            "for _ in range(5):\n"
            "    got, gs = capi.run_align(hmm, 'hipstr_hmm_', sb.ptr, fill=-3.25)\n"
            "    assert np.array_equal(gs, ws) and np.array_equal(got, want)\n"
            "print('ok')\n") % ROOT
    env = dict(os.environ, HIPSTR_WAIT=mode)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout


def test_one_locus_calls_from_many_host_threads(hmm, oracle, monkeypatch):
    """The unedited caller's shape (HapAligner::process_reads once per locus, README.md:167-171's N workers in ONE process): twelve host
    threads make one-locus calls at the same time — every thread has its own pair of streams (the main chain and, since round 5, the
    side stream of the device-built tables and the interrupted alleles' STR kernels), the block caches and the context are shared.  Loci with
    inherited interruptions (kernel _rp on the side stream), with random ones (_pw) and plain ones, each compared with the oracle bit for bit,
    three passes with the loci dealt differently so that a thread's streams see every kind after every other."""
    import threading
    loci = []
    for inherit, imperfect, kw in [(2, "0.0", dict(reads_per_locus=40, n_str_alleles=32)), (0, "1.0", dict(reads_per_locus=50, n_str_alleles=8)),
                                   (0, "0.05", dict(reads_per_locus=50, n_str_alleles=4)), (3, "0.3", dict(reads_per_locus=24, n_str_alleles=16, n_flank_opts=3))]:
        monkeypatch.setenv("HIPSTR_SYNTH_INHERIT", str(inherit))
        monkeypatch.setenv("HIPSTR_SYNTH_IMPERFECT", imperfect)
        for k in range(6):
            sb = capi.SynthBatch(n_loci=1, seed=900 + 10*len(loci) + k, **kw)
            want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25)
            loci.append((sb, want, ws))
    n_thr = 12
    bad = []
    def worker(t, rnd):
        try:
            for i in range(len(loci)):
                sb, want, ws = loci[(i*(2*rnd + 1) + 5*t) % len(loci)]
                got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
                if not (np.array_equal(gs, ws) and np.array_equal(got, want)):
                    bad.append((t, rnd, i))
        except Exception as ex:      # noqa: BLE001 - reported below
            bad.append((t, rnd, repr(ex)))
    for rnd in range(3):
        ths = [threading.Thread(target=worker, args=(t, rnd)) for t in range(n_thr)]
        for th in ths: th.start()
        for th in ths: th.join()
    assert not bad, bad[:5]


def test_api_profile_buckets(hmm):
    """hipstr_debug_api_profile: wall time and calls per entry point while switched on (what genotype_flow --profile prints)."""
    import ctypes as C
    hmm.hipstr_debug_api_profile.restype = C.c_int
    hmm.hipstr_debug_api_profile.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    sb = capi.SynthBatch(n_loci=2, reads_per_locus=20, n_str_alleles=6, seed=8)
    capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    names = (C.c_char_p * 32)(); secs = (C.c_double * 32)(); calls = (C.c_int64 * 32)()
    n = hmm.hipstr_debug_api_profile(1, 32, names, secs, calls)
    assert 8 <= n <= 32 and all(calls[i] == 0 for i in range(n))
    for _ in range(3):
        capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    hmm.hipstr_debug_api_profile(0, 32, names, secs, calls)
    by = {names[i].decode(): (secs[i], calls[i]) for i in range(n)}
    whole = by["hipstr_hmm_process_reads[_seeded]"]
    assert whole[1] == 3 and whole[0] > 0
    parts = sum(v[0] for k, v in by.items() if k.startswith("  ") and not k.startswith("    ") and v[1] == 3 and "replay" not in k)
    sub = sum(v[0] for k, v in by.items() if k.startswith("    upload:"))          # parts of the upload (the blocks are taken while the staging clock runs)
    assert 0 < sub <= 1.05 * (by["  pack staging buffer"][0] + by["  blocks + H2D enqueue"][0])
    assert 0.5 * whole[0] < parts <= 1.05 * whole[0]          # the indented buckets are the parts of the entry point above them
    capi.run_align(hmm, "hipstr_hmm_", sb.ptr)                # switched off: nothing is added
    hmm.hipstr_debug_api_profile(-1, 32, names, secs, calls)
    assert calls[0] == 3


def test_systolic_flank_kernels_match_the_shared_row_sweeps(hmm, oracle, monkeypatch):
    """Round 5: launches of a few flank items (a locus or two per call) give every (read side[, allele]) matrix a wavefront of its own —
    rows as lanes, anti-diagonal steps, wave_shr for the row above (hs_flank_systolic) — instead of the sweeps that put reads / alleles on
    the lanes and all rows of a flank on one workgroup.  Same cells, same operations: bit-identical to the oracle and to the other form,
    whichever is forced (HIPSTR_FLANK_SYSTOLIC = 0 never, 1 small launches, 2 every launch that fits 256 columns): flanks of 1 to 150
    rows (one row; more than one band of 64 rows), masked reads and alleles, several flank options, interrupted repeats."""
    cases = [dict(n_loci=2, reads_per_locus=40, n_str_alleles=32, seed=3),
             dict(n_loci=3, reads_per_locus=25, n_str_alleles=7, n_flank_opts=2, seed=9, mask_rate=0.2),
             dict(n_loci=2, reads_per_locus=30, n_str_alleles=5, flank_len=150, read_len=250, seed=4),
             dict(n_loci=2, reads_per_locus=30, n_str_alleles=6, flank_len=8, seed=5),
             dict(n_loci=40, reads_per_locus=60, n_str_alleles=16, seed=6)]
    for kw in cases:
        sb = capi.SynthBatch(**kw)
        want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-6.5)
        for mode in ("0", "1", "2"):
            monkeypatch.setenv("HIPSTR_FLANK_SYSTOLIC", mode)
            got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-6.5)
            assert np.array_equal(gs, ws) and np.array_equal(got, want), (kw, mode)
    monkeypatch.delenv("HIPSTR_FLANK_SYSTOLIC")
