#!/usr/bin/env python
"""Randomised parity of HETEROGENEOUS batches (GPU vs oracle): what the host pipeline hands the kernels in production is a batch of loci that
have nothing in common — read lengths, flank lengths, allele counts, periods, interrupted or plain repeats, masks — where the generator's
batches (and the other fuzzers') are loci of ONE shape.  2 ... 7 generator batches of different shapes are concatenated into one
hipstr_batch_t and go through process_reads in one call; and through the stream, one locus per submission, results in submission order.
usage: tools/fuzz_mixed.py [configs] [seed]"""
import os, sys
import ctypes as C
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from hipstr_amd import capi, shard


def arrays_from_ptr(bptr):
    """The numpy / bytes arrays (capi.Batch.arrays) behind a hipstr_batch_t*."""
    b = bptr.contents
    nl = b.n_loci
    arr = lambda p, n, dt=np.int32: np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt).copy()
    a = {}
    for k in ("blk_start", "blk_end", "blk_nopts"): a[k] = arr(getattr(b, k), 3 * nl)
    a["period"] = arr(b.period, nl); a["stutter"] = arr(b.stutter, 6 * nl, np.float64)
    nopt = int(a["blk_nopts"].sum())
    a["opt_off"] = arr(b.opt_off, nopt + 1)
    a["seq"] = C.string_at(b.seq, int(a["opt_off"][-1]))
    a["hap_off"] = arr(b.hap_off, nl + 1)
    a["realign_hap"] = None if not b.realign_hap else arr(b.realign_hap, int(a["hap_off"][-1]), np.uint8)
    a["read_off"] = arr(b.read_off, nl + 1); nr = int(a["read_off"][-1])
    a["base_off"] = arr(b.base_off, nr + 1); nb = int(a["base_off"][-1])
    a["bases"] = C.string_at(b.bases, nb); a["quals"] = C.string_at(b.quals, nb)
    a["read_start"] = arr(b.read_start, nr)
    a["cigar_off"] = arr(b.cigar_off, nr + 1); nc = int(a["cigar_off"][-1])
    a["cigar_op"] = C.string_at(b.cigar_op, nc); a["cigar_len"] = arr(b.cigar_len, nc)
    a["realign_read"] = None if not b.realign_read else arr(b.realign_read, nr, np.uint8)
    return a


def concat(parts):
    """One batch of all the loci of `parts` (arrays dicts), in order."""
    out = {}
    for k in ("blk_start", "blk_end", "blk_nopts", "period", "stutter", "read_start", "cigar_len"):
        out[k] = np.concatenate([p[k] for p in parts])
    def offs(k, base_of):
        acc, o = [np.zeros(1, np.int32)], 0
        for p in parts:
            acc.append((p[k][1:] + o).astype(np.int32)); o += int(p[k][-1])
        return np.concatenate(acc)
    for k in ("opt_off", "hap_off", "read_off", "base_off", "cigar_off"): out[k] = offs(k, None)
    for k in ("seq", "bases", "quals", "cigar_op"): out[k] = b"".join(bytes(p[k]) for p in parts) + b"\0"
    if any(p["realign_hap"] is not None for p in parts):
        out["realign_hap"] = np.concatenate([p["realign_hap"] if p["realign_hap"] is not None else np.ones(int(p["hap_off"][-1]), np.uint8) for p in parts])
    else: out["realign_hap"] = None
    if any(p["realign_read"] is not None for p in parts):
        out["realign_read"] = np.concatenate([p["realign_read"] if p["realign_read"] is not None else np.ones(int(p["read_off"][-1]), np.uint8) for p in parts])
    else: out["realign_read"] = None
    if out["cigar_len"].size == 0: out["cigar_len"] = np.zeros(1, np.int32)
    return out


def run(n_cfg, seed, hmm, ora, stream_every=3):
    rng = np.random.default_rng(seed)
    bad = 0; total = 0
    for c in range(n_cfg):
        parts = []
        for k in range(int(rng.integers(2, 8))):
            os.environ["HIPSTR_SYNTH_IMPERFECT"] = str(float(rng.choice([0.0, 0.05, 0.3, 1.0])))
            os.environ["HIPSTR_SYNTH_INHERIT"] = str(int(rng.choice([0, 0, 1, 2, 3])))
            kw = dict(n_loci=int(rng.integers(1, 3)), reads_per_locus=int(rng.choice([1, 3, 17, 40, 64, 65, 130])), n_str_alleles=int(rng.choice([1, 2, 5, 12, 32, 33, 70])),
                      read_len=int(rng.choice([12, 36, 75, 100, 150, 151, 250, 400])), flank_len=int(rng.choice([6, 20, 35, 60, 110, 200])), str_bp=int(rng.integers(4, 121)),
                      n_flank_opts=int(rng.choice([1, 1, 2, 3])), seed=int(rng.integers(1, 1 << 30)), mask_rate=float(rng.choice([0.0, 0.0, 0.3])))
            if kw["n_str_alleles"] * kw["n_flank_opts"] ** 2 > 300: kw["n_flank_opts"] = 1
            sb = capi.SynthBatch(**kw)
            parts.append(arrays_from_ptr(sb.ptr)); sb.close()
        b = shard.batch_from_arrays(concat(parts))
        want, ws = capi.run_align(ora, "oracle_", b.ptr, fill=-3.25)
        got, gs = capi.run_align(hmm, "hipstr_hmm_", b.ptr, fill=-3.25)
        total += got.size
        if not (np.array_equal(gs, ws) and np.array_equal(got, want)):
            bad += 1; print("MISMATCH one call, config", c, "loci", len(b.arrays["period"]), "n", got.size, "nbad", int((got != want).sum()), flush=True)
        if c % stream_every == 0:                  # the same loci through the stream, one locus per submission
            st = capi.Stream(hmm)
            st.submit_each(b.ptr)
            p2 = np.full(max(got.size, 1), -3.25); s2 = np.full(max(gs.size, 1), -7, np.int32)
            st.collect(len(b.arrays["period"]), p2, s2); st.close()
            if not (np.array_equal(s2[:gs.size], ws) and np.array_equal(p2[:got.size], want)):
                bad += 1; print("MISMATCH stream, config", c, flush=True)
    print("configs %d alignments %d mismatching %d" % (n_cfg, total, bad))
    return bad


if __name__ == "__main__":
    hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1, hmm, capi.load_oracle())
