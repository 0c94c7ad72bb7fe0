#!/usr/bin/env python
"""Randomised posterior / genotype-call parity at the kernels' size boundaries (diplotype counts around the 2048 a unit keeps in registers,
allele counts around a wavefront's 64 lanes and its multiples, samples without reads, weights 0, haploid loci, one variant ... as many as
haplotypes): hipstr_post_run and hipstr_post_extract against the oracle evaluated with the same correctly rounded exp / log
(oracle_set_cr_math: the level-2 contract of DESIGN §3), every output bit for bit.    usage: tools/fuzz_post.py [configs] [seed]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from hipstr_amd import capi

def run_post(hmm, pb):
    S = int(pb.samp_off[-1])
    post = np.zeros(max(int(pb.post_off[-1]), 1)); tot = np.zeros(max(S, 1)); gt = np.zeros(max(2 * S, 2), np.int32); ltot = np.zeros(max(pb.struct.n_loci, 1))
    rc = hmm.hipstr_post_run(pb.ptr, None, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p), gt.ctypes.data_as(capi._i32p), ltot.ctypes.data_as(capi._f64p))
    assert rc == 0, hmm.hipstr_last_error()
    return post[:int(pb.post_off[-1])], tot[:S], gt[:2 * S].reshape(-1, 2), ltot[:pb.struct.n_loci]

def run(n_cfg, seed, hmm, ora):
    rng = np.random.default_rng(seed)
    A_EDGE = [1, 2, 3, 5, 8, 31, 32, 33, 44, 45, 46, 47, 63, 64, 65, 90, 96, 127, 128, 129, 150]
    if os.environ.get("HIPSTR_FUZZ_BIG"):          # round 6: up to MAX_TOTAL_HAPLOTYPES = 1000 (genotyper_bam_processor.h:110): 10^6 diplotypes per sample
        A_EDGE = [150, 191, 192, 193, 255, 256, 257, 300, 383, 384, 500, 511, 512, 513, 640, 767, 768, 900, 999, 1000]
    bad = 0; n_dip = 0
    for c in range(n_cfg):
        nl = int(rng.integers(1, 6)) if not os.environ.get("HIPSTR_FUZZ_BIG") else int(rng.integers(1, 3))
        A, S, off, lab, hap, nv, h2a = [], [], [0], [], [], [], []
        for l in range(nl):
            a = int(rng.choice(A_EDGE)) if rng.random() < 0.8 else int(rng.integers(1, 80))
            s = int(rng.choice([1, 2, 3, 7, 40])) if a <= 65 else (int(rng.choice([1, 2, 5])) if a <= 200 else int(rng.choice([1, 2, 3])))
            reads = []
            for smp in range(s):
                k = int(rng.choice([0, 1, 2, 6, 9, 30], p=[.1, .15, .2, .35, .15, .05]))
                reads += [smp] * k
            A.append(a); S.append(s); lab += reads; off.append(off[-1] + len(reads)); hap.append(int(rng.random() < 0.3))
            v = int(rng.integers(1, min(a, 24 if a <= 200 else 300) + 1)); nv.append(v)
            m = rng.integers(0, v, size=a); m[:v] = rng.permutation(v); h2a += list(m)        # every variant has a haplotype
        n = off[-1]
        nll = int(np.dot(A, np.diff(off)))
        spread = float(rng.choice([1.0, 30.0, 300.0]))
        kw = dict(n_alleles=A, n_samples=S, read_off=np.array(off, np.int32), sample_label=np.array(lab, np.int32),
                  log_p1=-rng.random(n) * float(rng.choice([0.0, 1.0, 8.0])), log_p2=-rng.random(n) * float(rng.choice([0.0, 1.0, 8.0])),
                  read_weight=(rng.random(n) < 0.9).astype(np.int32), log_aln_probs=-rng.random(max(nll, 1))[:nll] * spread, haploid=hap)
        pb = capi.PostBatch(**kw)
        got = run_post(hmm, pb)
        ggt = capi.run_gt_extract(hmm, "hipstr_", pb, nv, h2a)
        with capi.oracle_cr_math(ora):
            want = capi.run_posteriors(ora, "oracle_", pb)
            wgt = capi.run_gt_extract(ora, "oracle_", pb, nv, h2a)
        ok = all(np.array_equal(a_, b_) for a_, b_ in zip(got, want))
        for k_ in wgt:
            if isinstance(wgt[k_], list): ok = ok and all(np.array_equal(x, y) for x, y in zip(ggt[k_], wgt[k_]))
            else: ok = ok and np.array_equal(ggt[k_], wgt[k_])
        n_dip += int(np.dot(np.array(A) ** 2, S))
        if not ok:
            bad += 1
            which = [i for i, (a_, b_) in enumerate(zip(got, want)) if not np.array_equal(a_, b_)]
            whichg = [k_ for k_ in wgt if (not all(np.array_equal(x, y) for x, y in zip(ggt[k_], wgt[k_]))) if isinstance(wgt[k_], list)] + \
                     [k_ for k_ in wgt if not isinstance(wgt[k_], list) and not np.array_equal(ggt[k_], wgt[k_])]
            print("MISMATCH config", c, "A", A, "S", S, "haploid", hap, "V", nv, "posterior outputs", which, "genotype outputs", whichg, flush=True)
    print("configs %d diplotype-units %d mismatching configs %d" % (n_cfg, n_dip, bad))
    return bad, n_dip


def main():
    hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1, hmm, capi.load_oracle())

if __name__ == "__main__":
    main()
