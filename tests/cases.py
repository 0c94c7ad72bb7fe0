"""Hand-built and seeded test batches.  Each entry is a function returning a finalized capi.Batch.
The same batches feed (a) tests/golden/make_golden.py (run through the compiled reference),
(b) the CPU tests of the oracle and the host logic, (c) the GPU parity tests."""
import numpy as np

from hipstr_amd import capi
from util import STUTTER, simple_locus, synth_to_batch

LF = "ACGTTGCATGCATGACCTGAGTCCATGACTTGACA"
RF = "TTGACCGTAGGCTAGGCTTAACGGATCCGATTAGC"


def kat_survey():
    """SURVEY.md §8(c) first known-answer vector."""
    b = capi.Batch()
    read = LF[10:] + "GATA" * 11 + RF[:25]
    b.add_locus([(100, 135, [LF]), (135, 175, ["GATA" * 10, "GATA" * 11, "GATA" * 12, "GATA" * 9]), (175, 210, [RF])], 4, STUTTER,
                [dict(seq=read, qual="I" * len(read), start=110, cigar=[("=", 65), ("I", 4), ("=", 25)])])
    return b.finalize()


def edge_reads():
    """Overhang on both sides, no-seed reads, N bases, out-of-range qualities, short reads, masked reads."""
    strs = ["CAG" * 8, "CAG" * 9, "CAG" * 7, "CAG" * 4 + "CAA" + "CAG" * 4, "CAG" * 12]
    hap = LF + strs[0] + RF
    n0 = len(LF)
    reads = []
    reads.append((hap[5:75], None, 5, True))                                     # plain spanning read
    reads.append(("GGTCA" + hap[:60], None, -5, True))                           # overhangs the haplotype on the left
    reads.append((hap[30:] + "ACGTACGTAC", None, 30, True))                      # overhangs on the right
    reads.append((strs[0][3:21], None, n0 + 3, True))                            # entirely inside the STR -> seed -1
    reads.append((hap[n0 - 4:n0 + 24 + 4], None, n0 - 4, True))                  # flanks too short for a seed -> -1
    reads.append((hap[10:40] + "N" + hap[41:80], None, 10, True))                # N base
    q = "".join(chr(33 + (i * 7) % 60) for i in range(70))                       # qualities beyond 'J'
    reads.append((hap[8:78], q, 8, True))
    q2 = "".join(" !\"#$%&"[i % 7] for i in range(66))                           # qualities at and below '!'
    reads.append((hap[12:78], q2, 12, True))
    reads.append((hap[n0 - 14:n0 + 6], None, n0 - 14, True))                     # short read ending inside the STR
    reads.append((hap[n0 + 20:n0 + 24 + 16], None, n0 + 20, True))               # short read starting inside the STR
    reads.append((hap[0:94], None, 0, False))                                    # not realigned: row must stay untouched
    mm = list(hap[3:93]); mm[20] = "A" if mm[20] != "A" else "C"; mm[70] = "T" if mm[70] != "T" else "G"
    reads.append(("".join(mm), None, 3, True))                                   # mismatches in both flanks
    ins = hap[2:30] + "T" + hap[30:90]                                           # 1-bp flank insertion
    reads.append((ins, None, 2, True, [("=", 28), ("I", 1), ("=", 60)]))
    dele = hap[2:30] + hap[32:92]                                                # 2-bp flank deletion
    reads.append((dele, None, 2, True, [("=", 28), ("D", 2), ("=", 60)]))
    stut = LF[5:] + "CAG" * 7 + RF[:30]                                          # -1 repeat stutter read
    reads.append((stut, None, 5, True, [("=", 30), ("D", 3), ("=", 21 + 30)]))
    b, _ = simple_locus(LF, strs, RF, 3, reads)
    return b.finalize()


def tiny_alleles():
    """STR alleles shorter than the period / than the largest deletions (num_deletions_ < 6), period 6 and 1."""
    b = capi.Batch()
    strs = ["ACGGTC" * 3, "ACGGTC" * 2, "ACGGTC", "ACGG", "AC", "ACGGTC" * 5]
    hap = LF + strs[0] + RF
    reads = [(hap[s:s + 70], None, s, True) for s in (0, 4, 9, 14)]
    reads.append((LF[3:] + strs[4] + RF[:33], None, 3, True, [("=", 32), ("D", 16), ("=", 2 + 33)]))
    simple_locus(LF, strs, RF, 6, reads, batch=b)
    homo = ["A" * 12, "A" * 11, "A" * 14, "A" * 3, "A"]
    lf2, rf2 = LF[:-1] + "C", "G" + RF[1:]
    hap2 = lf2 + homo[0] + rf2
    reads2 = [(hap2[s:s + 60], None, s, True) for s in (1, 6, 15, 20)]
    simple_locus(lf2, homo, rf2, 1, reads2, start=2000, batch=b)
    return b.finalize()


def boundary_homopolymers():
    """Flanks whose terminal runs continue into the STR block (cross-block homopolymer lengths) and
    long runs inside the flanks (indices up to MAX_HOMOP_LEN), alternative flanks of different lengths."""
    lf = "GATTACAGGCTTAACCCCCCCGTAGCATCGGAAAAAAAAAAAAAAAAAGT" + "TTT"
    rf = "TTTCGGATGGGGGGGGGGCATCAGTTACGGATCAAGCTA"
    strs = ["TTA" * 9, "TTA" * 10, "TTA" * 8, "TTA" * 11, "TAT" + "TTA" * 8]
    lf_alt = [lf[:20] + lf[21:], lf[:-1] + "A"]
    rf_alt = ["A" + rf[1:]]
    hap = lf + strs[0] + rf
    reads = [(hap[s:s + 100], None, s, True) for s in (0, 3, 10, 17)]
    reads.append((hap[25:110], "".join("F:,#"[i % 4] for i in range(85)), 25, True))
    b, A = simple_locus(lf, strs, rf, 3, reads, lf_opts=lf_alt, rf_opts=rf_alt)
    return b.finalize()


def homopolymer_two_copy():
    """A homopolymer locus (period 1) whose candidates include the TWO-copy allele next to flanks that continue its run: the one case in
    which Haplotype::homopolymer_length's cross-block extension goes past its first neighbour (Haplotype.cpp:262-270 with HapBlock's carried
    counter, HapBlock.cpp:7-30: the table entry of a one-run block is 2 (n - 1), equal to n for n = 2), so the leading-flank rows next to
    'GG' depend on the far flank too — and alleles that start with a run of three share first base and run with it.  Interrupted and pure
    alleles, three right and two left flank options (round 6: found by the fuzzers, tools/repro_r06_period1.py)."""
    pre, suf = "TCAGGATCCATGCATTACGATCAG", "CTGATCGTAATGCATGGATCCTGA"
    lfs = [pre + "ACGTTGCAGG", pre + "ACGTTGCATG"]; rfs = ["GGTACCATGC" + suf, "TGTACCATGC" + suf, "GTTACCATGC" + suf]
    strs = ["GGGGGG", "GG", "GGGAGGGG", "GGG", "GGGGCGGGG", "GGTGG", "G" * 9]
    hap = lfs[0] + strs[0] + rfs[0]
    reads = [(hap[s:s + ln], None, s, True) for s in (0, 2, 5, 9, 14) for ln in (40, 55, len(hap) - s) if s + ln <= len(hap)]
    alt = lfs[1] + strs[2] + rfs[1]
    reads += [(alt[3:60], "".join("F:,5"[i % 4] for i in range(57)), 3, True), (alt[8:], None, 8, True)]
    b, A = simple_locus(lfs[0], strs, rfs[0], 1, reads, lf_opts=lfs[1:], rf_opts=rfs[1:])
    assert A == 42
    return b.finalize()


def masks():
    """realign_to_haplotype / realign_read masks partially false on a multi-flank locus (2 x 4 x 2 = 16 alleles)."""
    strs = ["GT" * 12, "GT" * 11, "GT" * 13, "GT" * 6 + "GA" + "GT" * 5]
    lf_alt = [LF[:12] + "T" + LF[13:]]
    rf_alt = [RF[:20] + RF[21:]]
    hap = LF + strs[0] + RF
    reads = [(hap[s:s + 80], None, s, (s % 3) != 1) for s in (0, 2, 5, 7, 11, 13)]
    mask = [(k * 7 + 3) % 5 != 0 for k in range(16)]
    b, A = simple_locus(LF, strs, RF, 2, reads, lf_opts=lf_alt, rf_opts=rf_alt, realign_hap=mask)
    assert A == 16
    return b.finalize()


def empty_and_ragged():
    """A locus without reads between two ordinary loci (ragged batch)."""
    b = capi.Batch()
    strs = ["AGAT" * 6, "AGAT" * 7]
    hap = LF + strs[0] + RF
    simple_locus(LF, strs, RF, 4, [(hap[3:83], None, 3, True)], batch=b)
    simple_locus(LF, strs, RF, 4, [], start=900, batch=b)
    simple_locus(LF, ["AGAT" * 6], RF, 4, [(hap[6:86], None, 6, True), (hap[1:81], None, 1, True)], start=1300, batch=b)
    return b.finalize()


def _synth(**kw):
    return lambda: synth_to_batch(capi.SynthBatch(**kw))


CASES = {
    "kat_survey": kat_survey,
    "edge_reads": edge_reads,
    "tiny_alleles": tiny_alleles,
    "boundary_homopolymers": boundary_homopolymers,
    "homopolymer_two_copy": homopolymer_two_copy,
    "masks": masks,
    "empty_and_ragged": empty_and_ragged,
    # BASELINE.json configs[0]: 1 locus, 50 x 150 bp reads, 4 alleles
    "c1_plumbing": _synth(n_loci=1, reads_per_locus=50, n_str_alleles=4, seed=20260928),
    "synth_periods": _synth(n_loci=12, reads_per_locus=8, n_str_alleles=6, seed=7),
    "synth_multiflank": _synth(n_loci=4, reads_per_locus=8, n_str_alleles=4, n_flank_opts=3, seed=13, mask_rate=0.25),
    # configs[4]-like stress, shrunk: 250 bp reads, long STR blocks, deep matrices
    "c5_stress_small": _synth(n_loci=2, reads_per_locus=6, n_str_alleles=24, read_len=250, flank_len=110, str_bp=100, seed=5),
    "short_production_like": _synth(n_loci=4, reads_per_locus=10, n_str_alleles=8, read_len=100, flank_len=35, str_bp=30, seed=3),
    # round 6: 4 x 60 x 4 = 960 candidate haplotypes, next to MAX_TOTAL_HAPLOTYPES = 1000 (genotyper_bam_processor.h:110, enforced at
    # seq_stutter_genotyper.cpp:610-614); fixtures by make_golden.py's "sizes" section
    "many_haplotypes": _synth(n_loci=1, reads_per_locus=12, n_str_alleles=60, n_flank_opts=4, seed=11),
}

BIGPOST_STRIDE = 101


def thousand_haplotype_posteriors():
    """Posterior / genotype-call inputs with A = 1000 haplotypes (10^6 diplotypes per sample) for 3 samples, one without reads:
    (PostBatch kwargs, n_variants, hap_to_allele).  Likelihood rows shaped like alignments: a best haplotype, neighbours a few nats
    worse, the rest far off."""
    rng = np.random.default_rng(20261001)
    A, S = 1000, 3
    lab = np.array([0] * 7 + [2] * 5, np.int32); n = lab.size
    best = rng.integers(0, A, size=n)
    ll = -np.abs(np.arange(A)[None, :] - best[:, None]) * rng.uniform(0.05, 0.6, size=(n, 1)) - rng.random((n, A))
    p1 = np.where(rng.random(n) < 0.4, -rng.random(n) * 5, 0.0); p2 = np.where(p1 < 0, -rng.random(n) * 0.05, 0.0)
    w = np.ones(n, np.int32); w[3] = 0
    kw = dict(n_alleles=[A], n_samples=[S], read_off=[0, n], sample_label=lab, log_p1=p1, log_p2=p2, read_weight=w, log_aln_probs=ll.ravel(), haploid=[0])
    V = 250
    h2a = ((np.arange(A) // 2) % V).astype(np.int32)
    return kw, [V], h2a
