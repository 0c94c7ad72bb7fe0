"""Seeded inputs for the de novo stutter EM (EMStutterGenotyper): reads as observed STR sizes per sample."""
import numpy as np


def em_case(seed, n_loci=3, samples=(8, 30), reads_per_sample=(2, 9), haploid_rate=0.25, snp_rate=0.3):
    rng = np.random.default_rng(seed)
    period, n_samples, read_off, lab, bps, p1, p2, hap = [], [], [0], [], [], [], [], []
    for l in range(n_loci):
        p = int(rng.choice([2, 3, 4, 5, 6], p=[.35, .2, .3, .1, .05]))
        S = int(rng.integers(samples[0], samples[1] + 1))
        h = rng.random() < haploid_rate
        alleles = p * rng.integers(-4, 5, size=int(rng.integers(2, 6)))
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
