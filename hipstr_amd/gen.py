"""Seeded synthetic inputs for the stages around the HMM (shared by tests/ and bench.py): observed STR sizes per sample for the
de novo stutter EM, and (reference window, read) pairs for Needleman-Wunsch."""
import numpy as np


def em_case(seed, n_loci=3, samples=(8, 30), reads_per_sample=(2, 9), haploid_rate=0.25, snp_rate=0.3, allele_counts=None):
    """allele_counts: optional per-locus number of true alleles (default: 2..5 drawn per locus)."""
    rng = np.random.default_rng(seed)
    period, n_samples, read_off, lab, bps, p1, p2, hap = [], [], [0], [], [], [], [], []
    for l in range(n_loci):
        p = int(rng.choice([2, 3, 4, 5, 6], p=[.35, .2, .3, .1, .05]))
        S = int(rng.integers(samples[0], samples[1] + 1))
        h = rng.random() < haploid_rate
        if allele_counts is None:
            alleles = p * rng.integers(-4, 5, size=int(rng.integers(2, 6)))
        else:
            nal = int(allele_counts[l])
            alleles = p * rng.permutation(np.arange(-(nal // 2), nal - nal // 2))
        up, down, oof = rng.uniform(0.01, 0.08), rng.uniform(0.02, 0.12), rng.uniform(0.0, 0.02)
        n = 0
        for s in range(S):
            g = rng.choice(alleles, size=2)
            if h:
                g[1] = g[0]
            for _ in range(int(rng.integers(reads_per_sample[0], reads_per_sample[1] + 1))):
                strand = int(rng.integers(2))
                size = int(g[strand])
                u = rng.random()
                if u < up:
                    size += p * int(rng.geometric(0.85))
                elif u < up + down:
                    size -= p * int(rng.geometric(0.85))
                elif u < up + down + oof:
                    size += int(rng.choice([-1, 1])) * int(rng.geometric(0.8))
                lab.append(s); bps.append(size)
                if rng.random() < snp_rate and not h:
                    good, bad = -rng.random() * 0.05, -2 - rng.random() * 6
                    p1.append(good if strand == 0 else bad); p2.append(bad if strand == 0 else good)
                else:
                    p1.append(0.0); p2.append(0.0)
                n += 1
        period.append(p); n_samples.append(S); hap.append(1 if h else 0); read_off.append(read_off[-1] + n)
    return dict(period=period, n_samples=n_samples, read_off=read_off, sample_label=lab, num_bps=bps, log_p1=p1, log_p2=p2, haploid=hap)



def nw_pairs(seed, n=40, ref_len=(120, 320), read_len=(30, 150), repeats=True, ns=True):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L1 = int(rng.integers(ref_len[0], ref_len[1] + 1))
        ref = list(rng.choice(list("ACGT"), L1))
        if repeats and rng.random() < 0.7:                       # an STR inside the window: many equally good gap placements
            p = int(rng.integers(1, 7)); motif = list(rng.choice(list("ACGT"), p)); c = int(rng.integers(4, 16))
            at = int(rng.integers(10, max(11, L1 - p * c - 10)))
            ref[at:at + p * c] = motif * c
            ref = ref[:L1]
        L2 = int(rng.integers(read_len[0], min(read_len[1], L1 - 10) + 1))
        st = int(rng.integers(0, L1 - L2 + 1))
        read = ref[st:st + L2]
        i = 0
        while i < len(read):
            u = rng.random()
            if u < 0.01:
                read[i] = str(rng.choice(list("ACGT")))
            elif u < 0.02:
                del read[i:i + int(rng.integers(1, 9))]
            elif u < 0.03:
                read[i:i] = list(rng.choice(list("ACGT"), int(rng.integers(1, 9))))
            elif ns and u < 0.033:
                read[i] = "N"
            i += 1
        read = read[:max(256, read_len[1])]
        if len(read) == 0:
            read = ["A"]
        if ns and rng.random() < 0.1:
            ref[int(rng.integers(L1))] = "N"
        out.append(("".join(ref), "".join(read)))
    return out
