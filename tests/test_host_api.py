"""The C++ host API (include/hipstr_hmm.hpp): compiled here with g++, host-only parts on CPU, device parts on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_test")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "host_api_test.cpp")
    deps = [src, os.path.join(ROOT, "include", "hipstr_hmm.hpp"), os.path.join(ROOT, "include", "hipstr_hmm.h"),
            os.path.join(ROOT, "hipstr_amd", "csrc", "libhipstr_hmm.so")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        libdir = os.path.join(ROOT, "hipstr_amd", "csrc")
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                               "-L", libdir, "-lhipstr_hmm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def _run(*args):
    _build()
    out = subprocess.run([EXE] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert out.returncode == 0, out.stderr
    kv = {}
    for line in out.stdout.splitlines():
        k, _, v = line.partition(" ")
        kv.setdefault(k, []).append(v.split())
    return kv


def test_host_only_parts(_native_built):
    kv = _run()
    assert kv["num_combs"] == [["4"]]
    assert kv["num_pools"] == [["2"]]            # identical sequences pool together (read_pooler.cpp:3-20)
    assert kv["pool0_qual"] == [["F"]]           # upper median of {I,5,F} (base_quality.cpp:25)
    assert kv["seed"] == [["81", "84"]]          # seed 81 for the 65=4I25= read; 100= read: ties go right (HapAligner.cpp:246,259)


@pytest.mark.gpu
def test_device_parts(_native_built):
    import util
    (b, rr, aa, h2r, exp), = util.load_trace_fixture(os.path.join(ROOT, "tests", "golden", "trace_kat_survey.npz"))
    kv = _run("--gpu", *[s.decode() for s in h2r])
    # the C++ classes' traceback of the known-answer read against the golden vectors of the compiled reference
    rows = {int(r[0]): r[1:] for r in kv["trace"]}
    assert len(rows) == 4 and len(exp) >= 3
    for k, e in zip(aa, exp):
        got = rows[int(k)]
        assert got == [e["hap_aln"], str(e["stutter_size"]), e["str_seq"], e["flank_left"], e["flank_right"], str(e["flank_ins"]), str(e["flank_del"]),
                       str(e["aln_start"]), str(e["aln_stop"]), e["cigar"], e["aln_str"]]
    assert kv["trace_one"] == [[rows[1][0]]]
    # the mirror's process_read / trace_optimal_aln with a seed base of the caller's (7 bases left of calc_seed_base's) == the C-ABI's seeded calls
    import ctypes as C
    import numpy as np
    from hipstr_amd import capi as _capi
    hmm = _capi.load_hmm()
    assert int(rr[0]) == 0
    seed2 = np.array([81 - 7], np.int32); row = np.zeros(4); sd = np.zeros(1, np.int32)
    ptr = b.ptr
    assert hmm.hipstr_hmm_process_reads_seeded(ptr, seed2.ctypes.data_as(_capi._i32p), row.ctypes.data_as(_capi._f64p), sd.ctypes.data_as(_capi._i32p)) == 0
    assert [int(x, 16) for x in kv["seeded_row"][0]] == [int(v) for v in row.view(np.uint64)]
    best = int(np.argmax(row))
    tr = _capi.run_trace(hmm, "hipstr_hmm_", ptr, [0], [best], h2r, req_seed=[81 - 7])
    assert kv["seeded_trace"] == [[str(best), tr[0]["hap_aln"], tr[0]["hap_aln"]]]
    assert int(kv["seeded_fixed"][0][0], 16) == int(row.view(np.uint64)[2]) and kv["seeded_fixed"][0][1] == "-5.0"
    assert kv["aln_info_derived"] == [[x.decode() for x in h2r]]          # Haplotype::aln_haps_to_ref done by the library
    assert kv["trace_two"] == [[rows[2][7], rows[2][8], rows[2][9]]]
    assert kv["kat_seed"] == [["81"]]
    want = [-7.37582683338, -4.37708234692, -7.35198679536, -9.68433975817]
    assert all(abs(float(a) - b) < 5e-11 for a, b in zip(kv["kat_ll"][0], want))
    rows = {int(r[0]): (int(r[1]), [float(x) for x in r[2:]]) for r in kv["read_ll"]}
    # reads 0 and 1 are mates from the same pool: both rows = 2 x pool row; read 2 = pool row; all share the pool seed
    assert rows[0][1] == rows[1][1] and all(abs(a - 2 * b) < 1e-9 for a, b in zip(rows[0][1], rows[2][1]))
    assert rows[0][0] == rows[2][0] == 81 and rows[3][0] == 84
    assert abs(float(kv["post_total"][0][0]) - (-26.5207888808)) < 1e-9
    assert kv["post_gt"] == [["1", "0"], ["2", "2"]]
    assert all(abs(float(a) - b) < 1e-9 for a, b in zip(kv["post_first"][0], [-1.39627803547, -3.2336875797, -3.4679593086]))
    # Genotyper::extract_genotypes_and_likelihoods through the C++ class, checked against the oracle on the same posterior case
    from hipstr_amd import capi
    pb, _, _, _ = util.load_gt_fixture(os.path.join(ROOT, "tests", "golden", "gt_kat_survey.npz"))
    want = capi.run_gt_extract(capi.load_oracle(), "oracle_", pb, [2], [0, 1, 1])
    for row in kv["gt_call"]:
        s = int(row[0]); vals = [float(x) for x in row[1:]]
        exp = list(want["best_hap"][s]) + list(want["best_gt"][s]) + [want[k][s] for k in ("log_phased_post", "log_unphased_post", "hap_log_phased_post",
                                                                                          "hap_log_unphased_post", "gl_diff")]
        for g, p in zip(want["gls"][s], want["pls"][s]):
            exp += [g, p]
        exp += list(want["phased_gls"][s])
        assert len(vals) == len(exp) and all(abs(a - b) < 1e-9 * max(1, abs(b)) for a, b in zip(vals, exp))
    # EMStutterGenotyper through the C++ class against the oracle on the same reads
    sizes = [[0, 0, 4, 4, 0, -4, 4, 0], [8, 8, 8, 4, 8, 0], [-4, -4, -8, -4, 0, 0, 1], [0, 4, 0, 4, 8, 4, 0, -4]]
    lab, bps, p1, p2 = [], [], [], []
    for s, row in enumerate(sizes):
        for j, b in enumerate(row):
            lab.append(s); bps.append(b); p1.append(-0.02 if j % 3 == 0 else 0.0); p2.append(-3.5 if j % 3 == 0 else 0.0)
    tr, st, it, ll = capi.run_em(capi.load_oracle(), "oracle_", [4], [4], [0, len(lab)], lab, bps, p1, p2, haploid=[0])
    row = kv["em"][0]
    assert int(row[0]) == int(tr[0]) and int(row[1]) == int(it[0])
    assert all(abs(float(a) - b) < 1e-9 for a, b in zip(row[2:8], st[0])) and abs(float(row[8]) - ll[0]) < 1e-9 * max(1, abs(ll[0]))


# ---- ReadPooler / calc_hap_aln_probs of hipstr_hmm.hpp against the reference's classes (tests/golden/pool_scatter_*.npz)
import glob
import numpy as np

POOL = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "pool_scatter_*.npz")))


def _pool_case(path, tmp_path, scatter):
    import ctypes as C
    from hipstr_amd import capi
    import util
    d = np.load(path)
    b = util.batch_from_dict(d)
    lib = capi.load_hmm()
    lib.hipstr_batch_write.restype = C.c_int; lib.hipstr_batch_write.argtypes = [C.c_char_p, capi._BP]
    bpath = str(tmp_path / "case.hsb"); spath = str(tmp_path / "side.txt")
    assert lib.hipstr_batch_write(bpath.encode(), b.ptr) == 0
    with open(spath, "w") as f:
        for k in ("second_mate", "realign_pool", "copy_read"):
            f.write("%s %d %s\n" % (k, len(d[k]), " ".join(str(int(x)) for x in d[k])))
        f.write("prefill %d %s\n" % (len(d["prefill"]), " ".join("%016x" % x for x in d["prefill"].view(np.uint64))))
    kv = _run("--scatter" if scatter else "--pool", bpath, spath)
    assert int(kv["n_pools"][0][0]) == int(d["expect_n_pools"][0])
    assert [int(x) for x in kv["pool_index"][0]] == d["expect_pool_index"].tolist()
    off = d["expect_pool_qual_off"]; q = bytes(d["expect_pool_quals"])
    assert [r[0] for r in kv["pool_qual"]] == [q[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]
    return d, kv


@pytest.mark.parametrize("path", POOL, ids=[os.path.basename(p)[13:-4] for p in POOL])
def test_read_pooler_matches_reference(_native_built, tmp_path, path):
    """Pool index of every read and the per-position upper-median qualities of every pool (read_pooler.cpp:3-20, base_quality.cpp:11-28)."""
    assert len(POOL) >= 3
    _pool_case(path, tmp_path, scatter=False)


@pytest.mark.gpu
@pytest.mark.parametrize("path", POOL, ids=[os.path.basename(p)[13:-4] for p in POOL])
def test_calc_hap_aln_probs_matches_reference(_native_built, tmp_path, path):
    """The read-level matrix after the scatter and mate sums, bit for bit, incl. entries that must stay untouched
    (masked haplotypes, copy_read false) and seeds (seq_stutter_genotyper.cpp:519-568)."""
    d, kv = _pool_case(path, tmp_path, scatter=True)
    assert [int(x) for x in kv["seeds"][0]] == d["expect_seeds"].tolist()
    got = np.array([int(x, 16) for x in kv["ll"][0]], np.uint64)
    assert np.array_equal(got, d["expect_log_aln_probs"].view(np.uint64))
