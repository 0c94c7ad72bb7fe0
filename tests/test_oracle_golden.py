"""CPU: the oracle (oracle/hipstr_oracle.c) against the golden vectors produced by the compiled reference."""
import glob
import os

import numpy as np
import pytest

from hipstr_amd import capi
from util import batch_from_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALIGN = sorted(glob.glob(os.path.join(GOLD, "align_*.npz")))
POST = sorted(glob.glob(os.path.join(GOLD, "post_*.npz")))


def test_fixtures_present():
    assert len(ALIGN) >= 10 and len(POST) >= 5 and os.path.exists(os.path.join(GOLD, "scalars.npz"))


@pytest.mark.parametrize("path", ALIGN, ids=[os.path.basename(p)[6:-4] for p in ALIGN])
def test_align_bit_exact(oracle, path):
    d = np.load(path)
    b = batch_from_dict(d)
    sent = float(d["sentinel"][0])
    probs, seeds = capi.run_align(oracle, "oracle_", b.ptr, fill=sent)
    assert np.array_equal(seeds, d["expect_seeds"])
    assert np.array_equal(probs, d["expect_aln_probs"]), "max|diff| = %g" % np.max(np.abs(probs - d["expect_aln_probs"]))


def test_kat_values(oracle):
    """seed 81; LL = -7.37582683338, -4.37708234692, -7.35198679536, -9.68433975817."""
    d = np.load(os.path.join(GOLD, "align_kat_survey.npz"))
    probs, seeds = capi.run_align(oracle, "oracle_", batch_from_dict(d).ptr)
    assert seeds.tolist() == [81]
    assert np.allclose(probs, [-7.37582683338, -4.37708234692, -7.35198679536, -9.68433975817], rtol=0, atol=5e-12)


@pytest.mark.parametrize("path", POST, ids=[os.path.basename(p)[5:-4] for p in POST])
def test_posteriors_bit_exact(oracle, path):
    d = np.load(path)
    pb = capi.PostBatch(d["n_alleles"], d["n_samples"], d["read_off"], d["sample_label"], d["log_p1"], d["log_p2"], d["read_weight"],
                        d["log_aln_probs"], d["haploid"])
    post, tot, gt, ltot = capi.run_posteriors(oracle, "oracle_", pb)
    assert np.array_equal(post, d["expect_post"]) and np.array_equal(tot, d["expect_total"])
    assert np.array_equal(gt, d["expect_gt"]) and np.array_equal(ltot, d["expect_locus_total"])


def test_posterior_kat_values(oracle):
    """SURVEY.md §8(c) second vector: total -26.5207888808, MAP s1 1|0, s2 2|2."""
    d = np.load(os.path.join(GOLD, "post_kat_survey.npz"))
    assert abs(float(d["expect_locus_total"][0]) - (-26.5207888808)) < 5e-11
    assert d["expect_gt"].tolist() == [[1, 0], [2, 2]]
    assert np.allclose(d["expect_post"][:3], [-1.39627803547, -3.2336875797, -3.4679593086], rtol=0, atol=5e-11)


def test_scalar_tables(oracle):
    d = np.load(os.path.join(GOLD, "scalars.npz"))
    f64p = capi._f64p
    assert np.array_equal(d["int_log"], [oracle.oracle_int_log(i) for i in range(600)])
    assert np.array_equal(d["transition"], [[oracle.oracle_transition(w, h) for h in range(16)] for w in range(7)])
    assert np.array_equal(d["base_quality"], [[oracle.oracle_base_quality(q, c) for q in range(128)] for c in (0, 1)])
    params = d["pmf_params"]; pm = d["pmf"]; per_set = len(pm) // 2
    for i, row in enumerate(pm):
        sp = np.ascontiguousarray(params[i // per_set])
        assert oracle.oracle_stutter_pmf(sp.ctypes.data_as(f64p), int(row[0]), int(row[1]), int(row[2])) == row[3]
    for v, want in zip(d["lse_vec_in"], d["lse_vec_out"]):
        v = np.ascontiguousarray(v[~np.isnan(v)])
        assert oracle.oracle_fast_lse_vec(v.ctypes.data_as(f64p), len(v)) == want
    for (a, b), want in zip(d["lse2_in"], d["lse2_out"]):
        assert oracle.oracle_fast_lse2(a, b) == want
