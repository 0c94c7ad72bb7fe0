#!/usr/bin/env python
"""Randomised stutter-EM parity at the kernels' size boundaries (allele sizes around a wavefront's 64 lanes and its multiples, one sample ...
hundreds, samples with one read ... dozens, haploid loci, batches mixing all of these so that the compaction sees loci leave in every
order): hipstr_em_train against the oracle evaluated with the same correctly rounded exp / log, every output bit for bit (trained
flags, six parameters, iteration counts, final log-likelihood).    usage: tools/fuzz_em.py [configs] [seed]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from hipstr_amd import capi
from hipstr_amd.gen import em_case

def run(n_cfg, seed, hmm, ora):
    rng = np.random.default_rng(seed)
    bad = 0; n_loci = 0
    for c in range(n_cfg):
        nl = int(rng.integers(1, 9))
        shape = int(rng.integers(4))
        if shape == 0:   kw = em_case(int(rng.integers(1 << 30)), n_loci=nl, samples=(1, 6), reads_per_sample=(1, 40))
        elif shape == 1: kw = em_case(int(rng.integers(1 << 30)), n_loci=nl, samples=(60, 90), reads_per_sample=(2, 5), haploid_rate=0.0,
                                      allele_counts=[int(rng.choice([30, 60, 64, 66, 80])) for _ in range(nl)])
        elif shape == 2: kw = em_case(int(rng.integers(1 << 30)), n_loci=nl, samples=(200, 330), reads_per_sample=(1, 3), allele_counts=[int(rng.integers(2, 9)) for _ in range(nl)])
        else:            kw = em_case(int(rng.integers(1 << 30)), n_loci=nl, samples=(3, 50), reads_per_sample=(1, 12), haploid_rate=0.5, snp_rate=0.8)
        kw["max_iter"] = int(rng.choice([100, 100, 7, 1]))
        got = capi.run_em(hmm, "hipstr_", **kw)
        with capi.oracle_cr_math(ora):
            want = capi.run_em(ora, "oracle_", **kw)
        ok = all(np.array_equal(a, b) for a, b in zip(got, want))
        n_loci += nl
        sizes = [len(set(kw["num_bps"][kw["read_off"][l]:kw["read_off"][l + 1]])) for l in range(nl)]
        if not ok:
            bad += 1
            print("MISMATCH config", c, "shape", shape, "sizes", sizes, "samples", kw["n_samples"], "iters", list(got[2]), list(want[2]), flush=True)
    print("configs %d loci %d mismatching configs %d" % (n_cfg, n_loci, bad))
    return bad, n_loci


def main():
    hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 1, hmm, capi.load_oracle())

if __name__ == "__main__":
    main()
