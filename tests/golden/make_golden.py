"""Generates tests/golden/*.npz by running the cases of tests/cases.py through the COMPILED REFERENCE
(oracle/_ref/libhipstr_ref.so, built from /root/reference by oracle/Makefile).  Run in the build
container only:  python tests/golden/make_golden.py
Each fixture holds the inputs (flat batch arrays) and the reference's outputs; nothing of the
reference's source is stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hipstr_amd import capi   # noqa: E402
from cases import CASES       # noqa: E402
from util import batch_to_dict   # noqa: E402

SENTINEL = -12345.678


def seeded(ref):
    """process_read / trace_optimal_aln with seeds the CALLER chooses (HapAligner.h:83, :93) instead of calc_seed_base's."""
    import json
    from hipstr_amd import shard
    rng = np.random.default_rng(20260930)
    sb = capi.SynthBatch(n_loci=3, reads_per_locus=30, n_str_alleles=6, n_flank_opts=2, seed=5, mask_rate=0.1)
    from util import synth_to_batch
    b = synth_to_batch(sb)
    _, auto = capi.run_align(ref, "ref_", b.ptr, fill=SENTINEL)
    lens = np.diff(b.arrays["base_off"])
    seed_in = np.full(len(auto), -2, np.int32)
    for r in range(len(auto)):
        if auto[r] >= 0 and r % 3 != 2:            # two reads in three get a seed of the caller's: anywhere that leaves a base either side
            seed_in[r] = int(rng.integers(1, lens[r] - 1)) if r % 3 == 0 else int(np.clip(auto[r] + rng.integers(-9, 10), 1, lens[r] - 2))
        elif auto[r] == -7 or r % 11 == 5:
            seed_in[r] = -1 if r % 11 == 5 else -2
    probs, seeds = capi.run_align(ref, "ref_", b.ptr, fill=SENTINEL, seed_in=seed_in)
    d = batch_to_dict(b)
    d["seed_in"] = seed_in; d["expect_aln_probs"] = probs; d["expect_seeds"] = seeds; d["sentinel"] = np.array([SENTINEL])
    # tracebacks of locus 0 with the same seeds
    a = b.arrays
    one = shard.batch_from_arrays(shard.subset_arrays(a, 0, 1))
    A = int(a["hap_off"][1]); n0 = int(a["read_off"][1])
    rr = [r for r in range(n0) if seeds[r] >= 0 and (a["realign_read"] is None or a["realign_read"][r])][:16]
    aa = [int(rng.integers(A)) for _ in rr]
    exp = capi.run_trace(ref, "ref_", one.ptr, rr, aa, cap=1 << 20, req_seed=[int(seeds[r]) for r in rr])
    d["trace_read"] = np.array(rr, np.int32); d["trace_allele"] = np.array(aa, np.int32)
    d["trace_h2r"] = np.frombuffer(b"\n".join(capi.ref_hap_aln_info(ref, one.ptr, A)), dtype=np.uint8).copy()
    d["trace_expect"] = np.frombuffer(json.dumps(exp).encode(), dtype=np.uint8).copy()
    np.savez_compressed(os.path.join(HERE, "seeded_align_trace.npz"), **d)
    print("seeded: reads", len(seeds), "caller seeds", int((seed_in >= 0).sum()), "changed rows vs auto", int((seeds != auto).sum()), "traces", len(rr))


def _unpooled_case(seed, n_reads, n_str, n_flank, hap_mask_rate):
    """One locus whose reads repeat (same sequence, other qualities) and come partly in mate pairs: what SeqStutterGenotyper::init
    sees before pooling (seq_stutter_genotyper.cpp:490-511)."""
    from util import synth_to_batch
    rng = np.random.default_rng(seed)
    a = synth_to_batch(capi.SynthBatch(n_loci=1, reads_per_locus=n_reads, n_str_alleles=n_str, n_flank_opts=n_flank, seed=seed)).arrays
    nopt = a["blk_nopts"]; seqs = [bytes(a["seq"][a["opt_off"][i]:a["opt_off"][i + 1]]).decode() for i in range(int(nopt.sum()))]
    blocks, c = [], 0
    for k in range(3):
        blocks.append((int(a["blk_start"][k]), int(a["blk_end"][k]), seqs[c:c + nopt[k]])); c += nopt[k]
    reads, second_mate = [], []
    for r in range(n_reads):
        lo, hi = int(a["base_off"][r]), int(a["base_off"][r + 1])
        rd = dict(seq=bytes(a["bases"][lo:hi]).decode(), qual=bytes(a["quals"][lo:hi]).decode(), start=int(a["read_start"][r]),
                  cigar=[(chr(a["cigar_op"][i]), int(a["cigar_len"][i])) for i in range(a["cigar_off"][r], a["cigar_off"][r + 1])])
        copies = 1 + (rng.random() < 0.4) + (rng.random() < 0.15)
        for c in range(copies):                      # the same read seen again with other base qualities: one pool, median qualities
            q = rd["qual"] if c == 0 else "".join(rng.choice(list("#,:FI5"), p=[.05, .1, .2, .45, .1, .1]) for _ in rd["qual"])
            reads.append(dict(rd, qual=q)); second_mate.append(0)
    order = rng.permutation(len(reads)); reads = [reads[i] for i in order]
    for i in range(1, len(reads)):                   # mates follow each other and share a name (:499): mark some pairs
        if not second_mate[i - 1] and rng.random() < 0.2:
            second_mate[i] = 1
    A = int(np.prod(nopt))
    mask = [1] + [int(rng.random() >= hap_mask_rate) for _ in range(A - 1)] if hap_mask_rate > 0 else None
    b = capi.Batch(); b.add_locus(blocks, int(a["period"][0]), list(a["stutter"][:6]), reads, realign_hap=mask); b.finalize()
    return b, np.array(second_mate, np.uint8), A


def pool_scatter(ref):
    """ReadPooler (read_pooler.cpp:3-20, read_pooler.h:42-48) and SeqStutterGenotyper::calc_hap_aln_probs' scatter + mate sums
    (seq_stutter_genotyper.cpp:519-568) on the reference's classes: first round (everything realigned) and a later round (some
    haplotypes kept, some pools and reads skipped, old values in the matrix)."""
    import ctypes as C
    u8p = capi._u8p
    ref.ref_pool.restype = C.c_int; ref.ref_pool.argtypes = [capi._BP, capi._i32p, capi._i32p, C.c_char_p, capi._i32p, C.c_int32]
    ref.ref_pool_scatter.restype = C.c_int; ref.ref_pool_scatter.argtypes = [capi._BP, u8p, u8p, u8p, capi._f64p, capi._i32p]
    for name, (seed, n_reads, n_str, n_flank, hmask, partial) in dict(first_round=(11, 30, 5, 1, 0.0, False), later_round=(12, 36, 6, 2, 0.45, True),
                                                                       multiflank_all=(13, 24, 4, 2, 0.0, False)).items():
        rng = np.random.default_rng(seed)
        b, mates, A = _unpooled_case(seed, n_reads, n_str, n_flank, hmask)
        R = int(b.arrays["read_off"][1])
        pool_index = np.zeros(R, np.int32); n_pools = np.zeros(1, np.int32); cap = 1 << 20
        pq = C.create_string_buffer(cap); pqo = np.zeros(R + 1, np.int32)
        assert ref.ref_pool(b.ptr, pool_index.ctypes.data_as(capi._i32p), n_pools.ctypes.data_as(capi._i32p), pq, pqo.ctypes.data_as(capi._i32p), cap) == 0
        P = int(n_pools[0])
        realign_pool = (rng.random(P) > 0.3).astype(np.uint8) if partial else np.ones(P, np.uint8)
        copy_read = (rng.random(R) > 0.2).astype(np.uint8) if partial else np.ones(R, np.uint8)
        if partial:
            realign_pool[pool_index[copy_read == 1]] = 1       # the caller never copies from a pool it did not realign (:95-121 of assemble_flanks)
        ll = -rng.random(R * A) * 50 - 1; seeds = np.full(R, -9, np.int32)
        before = ll.copy()
        assert ref.ref_pool_scatter(b.ptr, mates.ctypes.data_as(u8p), realign_pool.ctypes.data_as(u8p), copy_read.ctypes.data_as(u8p),
                                    ll.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p)) == 0
        d = batch_to_dict(b)
        d.update(second_mate=mates, realign_pool=realign_pool, copy_read=copy_read, prefill=before, expect_pool_index=pool_index,
                 expect_n_pools=n_pools, expect_pool_quals=np.frombuffer(pq.raw[:pqo[P]], np.uint8).copy(), expect_pool_qual_off=pqo[:P + 1],
                 expect_log_aln_probs=ll, expect_seeds=seeds)
        np.savez_compressed(os.path.join(HERE, "pool_scatter_%s.npz" % name), **d)
        print("pool_scatter", name, "reads", R, "pools", P, "mates", int(mates.sum()), "alleles", A, "untouched", int((ll == before).sum()), "of", ll.size)


def sizes(ref):
    """Round 6: the sizes the reference allows and no earlier fixture reached — a locus with 4 x 60 x 4 = 960 candidate haplotypes
    (MAX_TOTAL_HAPLOTYPES = 1000, genotyper_bam_processor.h:110): forward log-likelihoods and tracebacks; posteriors and genotype calls
    with 1000 haplotypes (10^6 diplotypes per sample; a strided sample of the posteriors is stored, every other output whole)."""
    import json
    from hipstr_amd import shard
    from cases import CASES
    name = "many_haplotypes"
    b = CASES[name]()
    probs, seeds = capi.run_align(ref, "ref_", b.ptr, fill=SENTINEL)
    d = batch_to_dict(b)
    d["expect_aln_probs"] = probs; d["expect_seeds"] = seeds; d["sentinel"] = np.array([SENTINEL])
    np.savez_compressed(os.path.join(HERE, "align_%s.npz" % name), **d)
    a = b.arrays
    A = int(a["hap_off"][1])
    print(name, "alignments", probs.size, "haplotypes", A, "seed -1:", int((seeds == -1).sum()))
    rng_t = np.random.default_rng(20261001)
    ok = [r for r in range(int(a["read_off"][1])) if seeds[r] >= 0]
    rr, aa = [], []
    for r in ok:
        for k in list(rng_t.choice(A, size=5, replace=False)) + [0, A - 1]:
            rr.append(r); aa.append(int(k))
    exp = capi.run_trace(ref, "ref_", b.ptr, rr, aa, cap=1 << 22)
    h2r = capi.ref_hap_aln_info(ref, b.ptr, A)
    out = batch_to_dict(b, "L0_")
    out["L0_req_read"] = np.array(rr, np.int32); out["L0_req_allele"] = np.array(aa, np.int32)
    out["L0_h2r"] = np.frombuffer(b"\n".join(h2r), dtype=np.uint8).copy()
    out["L0_expect"] = np.frombuffer(json.dumps(exp).encode(), dtype=np.uint8).copy()
    out["n_traced"] = np.array([1])
    np.savez_compressed(os.path.join(HERE, "trace_%s.npz" % name), **out)
    print("trace", name, "requests", len(rr))
    # posteriors + calls at A = 1000
    from cases import thousand_haplotype_posteriors, BIGPOST_STRIDE
    kw, nv, h2a = thousand_haplotype_posteriors()
    pb = capi.PostBatch(**kw)
    post, tot, gt, ltot = capi.run_posteriors(ref, "ref_", pb)
    e = capi.run_gt_extract(ref, "ref_", pb, nv, h2a)
    out = dict(expect_post_strided=post[::BIGPOST_STRIDE].copy(), expect_post_max=np.array([post.max()]), expect_total=tot, expect_gt=gt, expect_locus_total=ltot)
    for k2 in ("best_hap", "best_gt", "log_phased_post", "log_unphased_post", "hap_log_phased_post", "hap_log_unphased_post", "gl_diff"):
        out["expect_" + k2] = e[k2]
    for k2 in ("gls", "pls", "phased_gls"):
        out["expect_" + k2] = np.concatenate(e[k2]) if e[k2] else np.zeros(0)
        out["expect_" + k2 + "_len"] = np.array([len(x) for x in e[k2]])
    np.savez_compressed(os.path.join(HERE, "bigpost_thousand_haplotypes.npz"), **out)
    print("bigpost: samples", tot.size, "posteriors", post.size, "stored", out["expect_post_strided"].size, "GLs", sum(len(x) for x in e["gls"]))


SECTIONS = {"seeded": seeded, "pool_scatter": pool_scatter, "sizes": sizes}


def main():
    ref = capi.load_ref()
    if len(sys.argv) > 2 and sys.argv[1] == "--only":       # python make_golden.py --only seeded [...]: just the named sections
        for name in sys.argv[2:]:
            SECTIONS[name](ref)
        return
    only = set(sys.argv[2:]) if len(sys.argv) > 2 and sys.argv[1] == "--cases" else None     # python make_golden.py --cases name [...]: the align_ / trace_ fixtures of these cases only
    for name, make in CASES.items():
        if name == "many_haplotypes":         # its own section ("sizes": other trace requests)
            continue
        if only is not None and name not in only:
            continue
        b = make()
        probs, seeds = capi.run_align(ref, "ref_", b.ptr, fill=SENTINEL)
        d = batch_to_dict(b)
        d["expect_aln_probs"] = probs; d["expect_seeds"] = seeds; d["sentinel"] = np.array([SENTINEL])
        np.savez_compressed(os.path.join(HERE, "align_%s.npz" % name), **d)
        print(name, "alignments", probs.size, "seed -1:", int((seeds == -1).sum()), "untouched:", int((probs == SENTINEL).sum()))

    # Viterbi traceback (HapAligner::trace_optimal_aln + stitch_alignment_trace) on one-locus cuts of the same cases
    import json
    from hipstr_amd import shard
    rng_t = np.random.default_rng(20260929)
    for name, make in CASES.items():
        if name == "many_haplotypes":
            continue
        if only is not None and name not in only:
            continue
        b = make()
        _, seeds = capi.run_align(ref, "ref_", b.ptr, fill=SENTINEL)
        a = b.arrays
        out = {}; traced = 0; n_req = 0
        for l in range(len(a["period"])):
            r0, r1 = int(a["read_off"][l]), int(a["read_off"][l + 1])
            A = int(a["hap_off"][l + 1] - a["hap_off"][l])
            ok = [r - r0 for r in range(r0, r1) if seeds[r] >= 0]
            if not ok or traced >= 6:
                continue
            one = shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1))
            rr, aa = [], []
            for r in ok[:12]:
                for k in rng_t.choice(A, size=min(A, 3), replace=False):
                    rr.append(r); aa.append(int(k))
            exp = capi.run_trace(ref, "ref_", one.ptr, rr, aa, cap=1 << 20)
            h2r = capi.ref_hap_aln_info(ref, one.ptr, A)
            pre = "L%d_" % traced
            out.update(batch_to_dict(one, pre))
            out[pre + "req_read"] = np.array(rr, np.int32); out[pre + "req_allele"] = np.array(aa, np.int32)
            out[pre + "h2r"] = np.frombuffer(b"\n".join(h2r), dtype=np.uint8).copy()
            out[pre + "expect"] = np.frombuffer(json.dumps(exp).encode(), dtype=np.uint8).copy()
            traced += 1; n_req += len(rr)
        if traced:
            out["n_traced"] = np.array([traced])
            np.savez_compressed(os.path.join(HERE, "trace_%s.npz" % name), **out)
        print("trace", name, "loci", traced, "requests", n_req)

    if only is not None:
        return
    # posteriors: SURVEY §8(c) second KAT + seeded random cases
    rng = np.random.default_rng(20260928)
    posts = {}
    LL = np.array([[-4.4, -7.3, -9.6], [-7.1, -4.2, -7.5], [-4.5, -7.0, -9.9], [-9.0, -6.0, -4.1], [-9.2, -6.3, -4.0]])
    posts["kat_survey"] = dict(n_alleles=[3], n_samples=[2], read_off=[0, 5], sample_label=[0, 0, 0, 1, 1], log_p1=[0, -0.01, 0, 0, 0],
                               log_p2=[0, -5, 0, 0, 0], read_weight=[1] * 5, log_aln_probs=LL.ravel(), haploid=[0])
    for t in range(4):
        nl = 5
        A = rng.integers(1, [4, 9, 17, 33][t], nl); S = rng.integers(1, 6, nl)
        R = [int(rng.integers(s, 8 * s + 1)) for s in S]
        ro = np.concatenate([[0], np.cumsum(R)])
        lab = np.concatenate([np.sort(rng.integers(0, s, r)) for s, r in zip(S, R)])
        n = int(ro[-1])
        posts["random_%d" % t] = dict(n_alleles=A, n_samples=S, read_off=ro, sample_label=lab,
                                      log_p1=-rng.random(n) * 3 * (rng.random(n) < 0.5), log_p2=-rng.random(n) * 3 * (rng.random(n) < 0.5),
                                      read_weight=(rng.random(n) < 0.85).astype(np.int32),
                                      log_aln_probs=np.concatenate([-rng.random(r * a) * 40 for r, a in zip(R, A)]),
                                      haploid=(rng.random(nl) < 0.3).astype(np.uint8))
    for name, kw in posts.items():
        pb = capi.PostBatch(**kw)
        post, tot, gt, ltot = capi.run_posteriors(ref, "ref_", pb)
        out = {k: np.asarray(v) for k, v in kw.items()}
        out.update(expect_post=post, expect_total=tot, expect_gt=gt, expect_locus_total=ltot)
        np.savez_compressed(os.path.join(HERE, "post_%s.npz" % name), **out)
        print("post", name, "samples", tot.size)

    # genotype calls (Genotyper::extract_genotypes_and_likelihoods) on the same posterior cases + a random haplotype->variant map
    rng_g = np.random.default_rng(20260930)
    for name, kw in posts.items():
        pb = capi.PostBatch(**kw)
        A = np.asarray(kw["n_alleles"])
        nv, h2a = [], []
        for a in A:
            v = int(rng_g.integers(1, a + 1))
            m = np.concatenate([np.arange(v), rng_g.integers(0, v, a - v)]); rng_g.shuffle(m)     # every variant carried by a haplotype
            nv.append(v); h2a.append(m)
        h2a = np.concatenate(h2a).astype(np.int32)
        e = capi.run_gt_extract(ref, "ref_", pb, nv, h2a)
        out = {k: np.asarray(v) for k, v in kw.items()}
        out.update(n_variants=np.array(nv, np.int32), hap_to_allele=h2a)
        for k2 in ("best_hap", "best_gt", "log_phased_post", "log_unphased_post", "hap_log_phased_post", "hap_log_unphased_post", "gl_diff"):
            out["expect_" + k2] = e[k2]
        for k2 in ("gls", "pls", "phased_gls"):
            out["expect_" + k2] = np.concatenate(e[k2]) if e[k2] else np.zeros(0)
            out["expect_" + k2 + "_len"] = np.array([len(x) for x in e[k2]])
        np.savez_compressed(os.path.join(HERE, "gt_%s.npz" % name), **out)
        print("gt", name, "samples", len(e["gl_diff"]), "GLs", sum(len(x) for x in e["gls"]))

    # de novo stutter EM (EMStutterGenotyper::train) on seeded length data
    from em_cases import em_case
    for name, kw in dict(small=em_case(1, n_loci=6), haploid_mix=em_case(2, n_loci=6, haploid_rate=0.6),
                         deep=em_case(3, n_loci=3, samples=(40, 60), reads_per_sample=(4, 10)),
                         no_snps=em_case(4, n_loci=5, snp_rate=0.0), few_iter=dict(em_case(5, n_loci=4), max_iter=3)).items():
        tr, st, it, ll = capi.run_em(ref, "ref_", **kw)
        out = {k: np.asarray(v) for k, v in kw.items()}
        out.update(expect_trained=tr, expect_stutter=st, expect_n_iter=it, expect_final_ll=ll)
        np.savez_compressed(os.path.join(HERE, "em_%s.npz" % name), **out)
        print("em", name, "loci", len(it), "iterations", list(it), "trained", list(tr.astype(int)))

    # Needleman-Wunsch (NeedlemanWunsch::Align), both stop rules
    import json
    from nw_cases import nw_pairs
    for name, (pairs, pen) in dict(reads_vs_window=(nw_pairs(1, n=40), False), haps_vs_ref=(nw_pairs(2, n=30, read_len=(80, 250)), True),
                                   short_and_n=(nw_pairs(3, n=40, ref_len=(20, 90), read_len=(1, 60)), False),
                                   no_repeats=(nw_pairs(4, n=20, repeats=False, ns=False), True)).items():
        e = capi.run_nw(ref, "ref_", pairs, pen)
        np.savez_compressed(os.path.join(HERE, "nw_%s.npz" % name), pairs=np.frombuffer(json.dumps(pairs).encode(), np.uint8).copy(),
                            end_penalty=np.array([int(pen)]), expect=np.frombuffer(json.dumps(e).encode(), np.uint8).copy())
        print("nw", name, "pairs", len(pairs), "with indels", sum(("I" in x[4] or "D" in x[4]) for x in e))

    # scalar probes: constant tables and the float log-sum-exp approximations
    f64p = capi._f64p
    vals = {}
    vals["int_log"] = np.array([ref.ref_int_log(i) for i in range(0, 600)])
    vals["transition"] = np.array([[ref.ref_transition(w, h) for h in range(16)] for w in range(7)])
    vals["base_quality"] = np.array([[ref.ref_base_quality(q, c) for q in range(128)] for c in (0, 1)])
    sp = np.array([0.9, 0.05, 0.05, 0.7, 0.005, 0.005]); sp2 = np.array([0.8, 0.1, 0.02, 0.6, 0.01, 0.02])
    pm = []
    for params in (sp, sp2):
        for period in (1, 2, 3, 4, 5, 6):
            for size in (0, 5, 24):
                for rd in range(max(0, size - 7 * period), size + 7 * period + 1):
                    pm.append([period, size, rd, ref.ref_stutter_pmf(params.ctypes.data_as(f64p), period, size, rd)])
    vals["pmf_params"] = np.stack([sp, sp2]); vals["pmf"] = np.array(pm)
    lse_in, lse_out = [], []
    for t in range(400):
        n = int(rng.integers(1, 40))
        v = -rng.random(n) * [2, 8, 30, 200][t % 4] - rng.random() * 50
        lse_in.append(np.pad(v, (0, 40 - n), constant_values=np.nan)); lse_out.append(ref.ref_fast_lse_vec(v.ctypes.data_as(f64p), n))
    vals["lse_vec_in"] = np.array(lse_in); vals["lse_vec_out"] = np.array(lse_out)
    ab = -rng.random((2000, 2)) * np.array([[1.0, 12.0]]) - rng.random((2000, 1)) * 30
    vals["lse2_in"] = ab; vals["lse2_out"] = np.array([ref.ref_fast_lse2(a, b) for a, b in ab])
    vals["log_thresh"] = np.array([ref.ref_log_thresh()]); vals["log_half"] = np.array([ref.ref_log_one_half()])
    np.savez_compressed(os.path.join(HERE, "scalars.npz"), **vals)
    print("scalars written")
    for fn in SECTIONS.values():
        fn(ref)


if __name__ == "__main__":
    main()
