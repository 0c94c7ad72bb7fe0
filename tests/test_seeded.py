"""Seeds chosen by the caller: HapAligner::process_read and trace_optimal_aln take the seed base as an ARGUMENT (HapAligner.h:83,
:93); only process_reads derives it with calc_seed_base.  The seeded entry points (hipstr_hmm_process_reads_seeded,
hipstr_hmm_trace_seeded) must therefore honour any seed that leaves a base on either side.  Golden vectors: the compiled
reference's process_read / trace_optimal_aln called with those seeds (tests/golden/make_golden.py::seeded)."""
import json
import os

import numpy as np
import pytest

from hipstr_amd import capi, shard
import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seeded_align_trace.npz")


def _load():
    d = np.load(GOLD)
    b = util.batch_from_dict(d)
    one = shard.batch_from_arrays(shard.subset_arrays(b.arrays, 0, 1))
    exp = json.loads(bytes(d["trace_expect"]).decode())
    for e in exp:
        e["indels"] = [tuple(x) for x in e["indels"]]; e["snps"] = [tuple(x) for x in e["snps"]]
    return d, b, one, exp


def _check(lib, prefix, tprefix):
    d, b, one, exp = _load()
    sent = float(d["sentinel"][0])
    probs, seeds = capi.run_align(lib, prefix, b.ptr, fill=sent, seed_in=d["seed_in"])
    assert np.array_equal(seeds, d["expect_seeds"]) and np.array_equal(probs, d["expect_aln_probs"])
    # the caller's seeds really were used: most rows differ from the calc_seed_base run
    auto, _ = capi.run_align(lib, prefix, b.ptr, fill=sent)
    assert (auto != probs).sum() > 0.2 * probs.size
    rr = d["trace_read"]; h2r = bytes(d["trace_h2r"]).split(b"\n")
    got = capi.run_trace(lib, tprefix, one.ptr, rr, d["trace_allele"], h2r, cap=1 << 20, req_seed=[int(d["expect_seeds"][r]) for r in rr])
    util.assert_traces_equal(got, exp, "seeded")


def test_oracle_matches_reference_vectors(oracle):
    _check(oracle, "oracle_", "oracle_")


@pytest.mark.gpu
def test_device_matches_reference_vectors(hmm):
    _check(hmm, "hipstr_hmm_", "hipstr_hmm_")


@pytest.mark.gpu
def test_seed_must_leave_a_base_on_either_side(hmm):
    d, b, _, _ = _load()
    bad = d["seed_in"].copy(); bad[0] = 0
    with pytest.raises(RuntimeError):
        capi.run_align(hmm, "hipstr_hmm_", b.ptr, seed_in=bad)
    assert b"either side" in hmm.hipstr_last_error()


@pytest.mark.gpu
def test_extreme_seeds_against_the_oracle(hmm, oracle):
    """Seeds one base from either read end (sides of 1 and len-2 columns), mixed with ordinary ones: read sides of very different
    lengths in one locus — the STR groups then pack a dozen one-column sides next to full-length ones."""
    sb = capi.SynthBatch(n_loci=4, reads_per_locus=48, n_str_alleles=16, seed=77)
    lens = np.diff(np.ctypeslib.as_array(sb.ptr.contents.base_off, shape=(sb.n_reads + 1,)))
    rng = np.random.default_rng(7)
    seed_in = np.full(sb.n_reads, -2, np.int32)                       # -2: compute (HIPSTR_SEED_AUTO)
    pick = rng.random(sb.n_reads)
    seed_in[pick < 0.25] = 1
    hi = (pick >= 0.25) & (pick < 0.5)
    seed_in[hi] = lens[hi] - 2
    mid = (pick >= 0.5) & (pick < 0.75)
    seed_in[mid] = rng.integers(2, lens[mid] - 2)
    want, ws = capi.run_align(oracle, "oracle_", sb.ptr, fill=-3.25, seed_in=seed_in)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25, seed_in=seed_in)
    assert np.array_equal(gs, ws) and np.array_equal(got, want)
