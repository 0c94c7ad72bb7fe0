#!/usr/bin/env python
"""Randomised parity of the stages around the forward path (GPU vs oracle): Needleman-Wunsch on pairs whose lengths sit around the kernel's
tile sizes (1 ... 1536-base reads against 1 ... 3000-base windows), the forward path and the traceback with seeds chosen by the caller
anywhere in the read (process_read's / trace_optimal_aln's seed_base argument: sides of 1 ... len-2 columns).
usage: tools/fuzz_misc.py [configs] [seed]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from hipstr_amd import capi


def nw_pairs(rng, n, ref_len, read_len, repeats, ns):
    """(reference window, read) pairs: the read a mutated piece of the window when it fits, random bases when it is longer than the window."""
    out = []
    for _ in range(n):
        L1 = int(rng.integers(ref_len[0], ref_len[1] + 1)); L2 = int(rng.integers(read_len[0], read_len[1] + 1))
        ref = list(rng.choice(list("ACGT"), L1))
        if repeats and L1 > 40 and rng.random() < 0.7:
            p = int(rng.integers(1, 7)); c = int(rng.integers(4, 16)); at = int(rng.integers(0, max(1, L1 - p * c)))
            ref[at:at + p * c] = list(rng.choice(list("ACGT"), p)) * c; ref = ref[:L1]
        if L2 <= L1:
            st = int(rng.integers(0, L1 - L2 + 1)); read = ref[st:st + L2]
            i = 0
            while i < len(read):
                u = rng.random()
                if u < 0.01: read[i] = str(rng.choice(list("ACGT")))
                elif u < 0.02 and len(read) > 4: del read[i:i + int(rng.integers(1, 4))]
                elif u < 0.03: read[i:i] = list(rng.choice(list("ACGT"), int(rng.integers(1, 4))))
                elif ns and u < 0.033: read[i] = "N"
                i += 1
        else:
            read = list(rng.choice(list("ACGT"), L2))
        read = read[:1536] or ["A"]
        out.append(("".join(ref), "".join(read)))
    return out


def run(n_cfg, seed, hmm, ora):
    rng = np.random.default_rng(seed)
    bad = 0; n_pairs = 0; n_aln = 0; n_tr = 0
    EDGE = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1535, 1536]
    for c in range(n_cfg):
        # --- Needleman-Wunsch
        rl = int(rng.choice(EDGE)); fl = int(rng.choice(EDGE + [2000, 3000]))
        kw = dict(n=int(rng.integers(1, 24)), read_len=(max(1, rl - 2), min(1536, rl + 2)), ref_len=(max(1, fl - 2), fl + 2),
                  repeats=bool(rng.integers(2)), ns=bool(rng.integers(2)))
        pairs = nw_pairs(rng, **kw); pen = bool(rng.integers(2))
        n_pairs += len(pairs)
        if capi.run_nw(hmm, "hipstr_", pairs, pen) != capi.run_nw(ora, "oracle_", pairs, pen):
            bad += 1; print("MISMATCH nw", kw, pen, flush=True)
        # --- caller-chosen seeds, forward and traceback
        os.environ["HIPSTR_SYNTH_IMPERFECT"] = str(float(rng.choice([0.0, 0.05, 1.0]))); os.environ["HIPSTR_SYNTH_INHERIT"] = str(int(rng.choice([0, 0, 2])))
        skw = dict(n_loci=1, reads_per_locus=int(rng.integers(1, 70)), n_str_alleles=int(rng.integers(1, 34)), read_len=int(rng.integers(12, 251)),
                   flank_len=int(rng.integers(6, 120)), str_bp=int(rng.integers(4, 100)), n_flank_opts=int(rng.integers(1, 3)), seed=int(rng.integers(1, 1 << 30)))
        sb = capi.SynthBatch(**skw)
        b = sb.ptr.contents
        base_off = np.ctypeslib.as_array(b.base_off, shape=(sb.n_reads + 1,))
        lens = np.diff(base_off)
        seeds_in = np.array([int(rng.integers(1, max(2, L - 1))) if rng.random() < 0.8 else -2 for L in lens], np.int32)     # -2: calc_seed_base
        seeds_in[lens < 3] = -2
        try:
            want, ws = capi.run_align(ora, "oracle_", sb.ptr, fill=-3.25, seed_in=seeds_in)
        except RuntimeError as e:
            print("oracle refused", skw, str(e)[:60]); continue
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25, seed_in=seeds_in)
        n_aln += got.size
        if not (np.array_equal(gs, ws) and np.array_equal(got, want)):
            bad += 1; print("MISMATCH seeded forward", skw, flush=True)
        A = sb.n_out // sb.n_reads
        rr = [r for r in range(sb.n_reads) if ws[r] >= 0][:40]; aa = [int(rng.integers(A)) for _ in rr]
        if rr:
            h2r = capi.hap_aln_info(ora, "oracle_", sb.ptr)
            sd = [int(ws[r]) for r in rr]
            wt = capi.run_trace(ora, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 21, req_seed=sd)
            gt = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 21, req_seed=sd)
            n_tr += len(rr)
            if gt != wt:
                bad += 1; print("MISMATCH seeded traceback", skw, flush=True)
    print("configs %d nw pairs %d seeded alignments %d seeded tracebacks %d mismatching %d" % (n_cfg, n_pairs, n_aln, n_tr, bad))
    return bad


if __name__ == "__main__":
    hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1, hmm, capi.load_oracle())
