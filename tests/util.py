"""Helpers shared by the tests: hand-built loci, batch <-> npz conversion."""
import numpy as np

from hipstr_amd import capi

STUTTER = [0.9, 0.05, 0.05, 0.7, 0.005, 0.005]

_ARRAY_KEYS = ["blk_start", "blk_end", "blk_nopts", "period", "stutter", "opt_off", "hap_off", "read_off", "base_off",
               "read_start", "cigar_off", "cigar_len"]
_BYTES_KEYS = ["seq", "bases", "quals", "cigar_op"]
_MASK_KEYS = ["realign_hap", "realign_read"]


def batch_to_dict(b, prefix=""):
    """capi.Batch -> dict of numpy arrays (for np.savez)."""
    out = {}
    for k in _ARRAY_KEYS:
        out[prefix + k] = b.arrays[k]
    for k in _BYTES_KEYS:
        out[prefix + k] = np.frombuffer(b.arrays[k], dtype=np.uint8).copy()
    for k in _MASK_KEYS:
        out[prefix + k] = b.arrays[k] if b.arrays[k] is not None else np.zeros(0, np.uint8)
    return out


def batch_from_dict(d, prefix=""):
    """Rebuild a capi.Batch (struct + arrays) from the dict written by batch_to_dict."""
    import ctypes as C
    b = capi.Batch()
    a = {}
    for k in _ARRAY_KEYS:
        a[k] = np.ascontiguousarray(d[prefix + k])
    for k in _BYTES_KEYS:
        a[k] = bytes(np.asarray(d[prefix + k], dtype=np.uint8).tobytes())
    for k in _MASK_KEYS:
        m = np.asarray(d[prefix + k], dtype=np.uint8)
        a[k] = np.ascontiguousarray(m) if m.size else None
    b.arrays = a
    s = capi.HipstrBatch()
    s.n_loci = len(a["period"])
    for k in ("blk_start", "blk_end", "blk_nopts", "period", "opt_off", "hap_off", "read_off", "base_off", "read_start", "cigar_off", "cigar_len"):
        setattr(s, k, a[k].ctypes.data_as(capi._i32p))
    s.stutter = a["stutter"].ctypes.data_as(capi._f64p)
    s.seq, s.bases, s.quals, s.cigar_op = a["seq"], a["bases"], a["quals"], a["cigar_op"]
    s.realign_hap = None if a["realign_hap"] is None else a["realign_hap"].ctypes.data_as(capi._u8p)
    s.realign_read = None if a["realign_read"] is None else a["realign_read"].ctypes.data_as(capi._u8p)
    s._keepalive = a     # byref(struct) keeps the struct alive; the struct keeps the arrays alive
    b.struct = s
    return b


def synth_to_batch(sb):
    """Copy a SynthBatch (C++ generator) into a capi.Batch so it can be stored as a fixture."""
    import ctypes as C
    p = sb.ptr.contents
    n = p.n_loci
    nblk = 3 * n
    as_i32 = lambda ptr, cnt: np.ctypeslib.as_array(ptr, shape=(cnt,)).copy() if cnt else np.zeros(0, np.int32)
    d = {}
    d["blk_start"] = as_i32(p.blk_start, nblk); d["blk_end"] = as_i32(p.blk_end, nblk); d["blk_nopts"] = as_i32(p.blk_nopts, nblk)
    d["period"] = as_i32(p.period, n)
    d["stutter"] = np.ctypeslib.as_array(p.stutter, shape=(6 * n,)).copy()
    nopt = int(d["blk_nopts"].sum())
    d["opt_off"] = as_i32(p.opt_off, nopt + 1)
    d["seq"] = np.frombuffer(C.string_at(p.seq, int(d["opt_off"][-1])) + b"\0", dtype=np.uint8).copy()
    d["hap_off"] = as_i32(p.hap_off, n + 1); d["read_off"] = as_i32(p.read_off, n + 1)
    nr = int(d["read_off"][-1])
    d["base_off"] = as_i32(p.base_off, nr + 1)
    nb = int(d["base_off"][-1])
    d["bases"] = np.frombuffer(C.string_at(p.bases, nb) + b"\0", dtype=np.uint8).copy()
    d["quals"] = np.frombuffer(C.string_at(p.quals, nb) + b"\0", dtype=np.uint8).copy()
    d["read_start"] = as_i32(p.read_start, nr); d["cigar_off"] = as_i32(p.cigar_off, nr + 1)
    nc = int(d["cigar_off"][-1])
    d["cigar_op"] = np.frombuffer(C.string_at(p.cigar_op, nc) + b"\0", dtype=np.uint8).copy()
    d["cigar_len"] = as_i32(p.cigar_len, nc) if nc else np.zeros(1, np.int32)
    A = int(d["hap_off"][-1])
    d["realign_hap"] = np.ctypeslib.as_array(p.realign_hap, shape=(A,)).copy() if p.realign_hap else np.zeros(0, np.uint8)
    d["realign_read"] = np.ctypeslib.as_array(p.realign_read, shape=(nr,)).copy() if p.realign_read else np.zeros(0, np.uint8)
    return batch_from_dict(d)


def cigar_vs_ref(read, hap_ref, offset):
    """'='/'X' CIGAR of `read` laid gap-free on hap_ref starting at hap_ref[offset] (bases outside count as '=')."""
    ops = []
    for i, c in enumerate(read):
        j = offset + i
        op = "=" if (j < 0 or j >= len(hap_ref) or hap_ref[j] == c) else "X"
        if ops and ops[-1][0] == op:
            ops[-1][1] += 1
        else:
            ops.append([op, 1])
    return [(o, n) for o, n in ops]


def simple_locus(lf, str_opts, rf, period, reads, start=500, lf_opts=None, rf_opts=None, realign_hap=None, batch=None):
    """reads: list of (sequence, quals or None, offset into the reference haplotype, realign flag[, cigar])."""
    b = batch if batch is not None else capi.Batch()
    ref_hap = lf + str_opts[0] + rf
    rds = []
    for rd in reads:
        seq, qual, off, flag = rd[:4]
        cig = rd[4] if len(rd) > 4 else cigar_vs_ref(seq, ref_hap, off)
        rds.append(dict(seq=seq, qual=qual if qual is not None else "F" * len(seq), start=start + off, cigar=cig, realign=flag))
    blocks = [(start, start + len(lf), [lf] + (lf_opts or [])),
              (start + len(lf), start + len(lf) + len(str_opts[0]), list(str_opts)),
              (start + len(lf) + len(str_opts[0]), start + len(ref_hap), [rf] + (rf_opts or []))]
    A = b.add_locus(blocks, period, STUTTER, rds, realign_hap=realign_hap)
    return b, A


def synthetic_hap_to_ref(ora, bptr):
    """A syntactically valid Haplotype::get_aln_info()-style string per allele of a ONE-locus batch: every block is
    aligned to option 0 of the same block end to end (common prefix 'M', then 'I' or 'D' for the length difference).
    stitch_alignment_trace only needs #M + #I == haplotype length; real strings come with the golden fixtures."""
    import ctypes as C
    b = bptr.contents if hasattr(bptr, "contents") else (bptr._obj if hasattr(bptr, "_obj") else bptr)
    nopts = np.ctypeslib.as_array(b.blk_nopts, shape=(3,)).astype(np.int32)
    opt_off = np.ctypeslib.as_array(b.opt_off, shape=(int(nopts.sum()) + 1,))
    lens, cur = [], 0
    for k in range(3):
        lens.append([int(opt_off[cur + o + 1] - opt_off[cur + o]) for o in range(nopts[k])])
        cur += int(nopts[k])
    out = []
    opts = np.zeros(3, np.int32)
    i32p = C.POINTER(C.c_int32)
    for k in range(int(np.prod(nopts))):
        ora.oracle_allele_options(nopts.ctypes.data_as(i32p), k, opts.ctypes.data_as(i32p))
        s = ""
        for blk in range(3):
            n, n0 = lens[blk][opts[blk]], lens[blk][0]
            s += "M" * min(n, n0) + ("I" * (n - n0) if n > n0 else "D" * (n0 - n))
        out.append(s.encode())
    return out


TRACE_FIELDS = ("max_index", "hap_aln", "stutter_size", "str_seq", "flank_left", "flank_right", "flank_ins", "flank_del", "indels", "snps",
                "aln_start", "aln_stop", "cigar", "aln_str")


def load_trace_fixture(path):
    """tests/golden/trace_*.npz -> list of (one-locus capi.Batch, req_read, req_allele, hap_to_ref, expected dicts)."""
    import json
    d = np.load(path)
    out = []
    for i in range(int(d["n_traced"][0])):
        pre = "L%d_" % i
        b = batch_from_dict(d, pre)
        h2r = bytes(d[pre + "h2r"].tobytes()).split(b"\n")
        exp = json.loads(bytes(d[pre + "expect"].tobytes()).decode())
        for e in exp:     # json turned the tuples into lists
            e["indels"] = [tuple(x) for x in e["indels"]]; e["snps"] = [tuple(x) for x in e["snps"]]
        out.append((b, d[pre + "req_read"], d[pre + "req_allele"], h2r, exp))
    return out


def assert_traces_equal(got, want, what=""):
    assert len(got) == len(want)
    for q, (g, w) in enumerate(zip(got, want)):
        assert g["ll"] == w["ll"], "%s request %d: ll %r != %r" % (what, q, g["ll"], w["ll"])
        for f in TRACE_FIELDS:
            if f == "max_index" and w[f] == -1:       # the reference keeps it in a local; the seed's 'M' in hap_aln pins it
                continue
            assert g[f] == w[f], "%s request %d: %s %r != %r" % (what, q, f, g[f], w[f])


def load_gt_fixture(path):
    """tests/golden/gt_*.npz -> (PostBatch, n_variants, hap_to_allele, expected dict in run_gt_extract's shape)."""
    d = np.load(path)
    pb = capi.PostBatch(d["n_alleles"], d["n_samples"], d["read_off"], d["sample_label"], d["log_p1"], d["log_p2"], d["read_weight"],
                        d["log_aln_probs"], d["haploid"])
    exp = {k: d["expect_" + k] for k in ("best_hap", "best_gt", "log_phased_post", "log_unphased_post", "hap_log_phased_post",
                                         "hap_log_unphased_post", "gl_diff")}
    for k in ("gls", "pls", "phased_gls"):
        cuts = np.cumsum(d["expect_" + k + "_len"])[:-1]
        exp[k] = np.split(d["expect_" + k], cuts)
    return pb, d["n_variants"], d["hap_to_allele"], exp


FLOAT_STEP = 3e-6
FLOAT_STEPS_SEEN = [0, 0]      # over the session: values that needed the widened window, values compared
BOUNDARY_REL = 1e-11           # how close the reference's cast argument must sit to a float rounding boundary for a step to be owed, relative to the larger of the two log-sum-exp arguments (>= 1)


def float_boundary_distance(delta):
    """Distance of the double `delta` — the argument the reference's fast_log_sum_exp(a, b) casts to float (mathops.cpp:86-95:
    delta = min - max) — from the nearest point where (float)delta changes, i.e. the midpoint between two adjacent floats, or from
    LOG_THRESH where the function switches branch."""
    import math
    f = np.float32(delta)
    lo = np.nextafter(f, np.float32(-np.inf)); hi = np.nextafter(f, np.float32(np.inf))
    m_lo = 0.5 * (float(f) + float(lo)); m_hi = 0.5 * (float(f) + float(hi))
    return min(abs(delta - m_lo), abs(m_hi - delta), abs(delta - math.log(0.001)))


def _genotype_totals(post, A, V, h2a):
    """T[v1, v2] = log-sum-exp of the posteriors of the haplotype pairs that carry the STR variants (v1, v2) (genotyper.cpp:150-170)."""
    T = np.full((V, V), -np.inf)
    P = post.reshape(A, A)
    members = [np.nonzero(np.asarray(h2a) == v)[0] for v in range(V)]
    for v1 in range(V):
        for v2 in range(V):
            if len(members[v1]) and len(members[v2]):
                x = P[np.ix_(members[v1], members[v2])].ravel()
                m = x.max(); T[v1, v2] = m + np.log(np.exp(x - m).sum())
    return T


def assert_genotypes_close(got, want, tol, what="", verify=None):
    """tol = 0 demands identical bits.  With tol > 0 (device exp/log in the exact log-sum-exps: posteriors and per-genotype totals carry
    ~1e-13 of rounding noise):
      * values are compared with |d| <= tol * max(1, |x|);
      * the values that pass through the reference's FLOAT pair log-sum-exp (fast_log_sum_exp(a, b), mathops.cpp:86-95:
        hap_log_unphased_post, every GL, hence GLDIFF and PL) may in addition sit one float rounding step away: that function casts the
        difference of its arguments to float and runs bit-trick exp/log on it, so noise of 1e-13 in an argument flips the float rounding
        where — and only where — the argument sits on a rounding boundary, and moves the result by up to 2^-17 ln 2 / 2 = 2.7e-6 nats.
        Such steps must be <= FLOAT_STEP, rare (<= 0.5 % of the compared values; observed 0.2 %), and — with verify = (oracle library, PostBatch,
        n_variants, hap_to_allele) — OWED: the test recomputes the argument of the reference's cast from the oracle's posteriors and
        requires it to lie within BOUNDARY_REL = 1e-11 (relative to the arguments' magnitude) of a float rounding boundary or of LOG_THRESH.  A GL that differs
        without such a boundary is a failure, however small the difference;
      * GLDIFF may differ where one of the sample's GLs made an owed step; a PL (a truncated integer of -10 (GL - maxGL)) may differ by
        one where that product sits within 10 FLOAT_STEP of an integer."""
    assert np.array_equal(got["best_hap"], want["best_hap"]), what
    assert np.array_equal(got["best_gt"], want["best_gt"]), what
    steps = [0, 0]          # float steps seen, values compared
    stepped = []            # (kind, sample, index) of every value outside tol
    def close(a, b, float_lse=False, nstep=1, kind=None, sample=None):
        a = np.asarray(a, float); b = np.asarray(b, float)
        if tol == 0:
            return np.array_equal(a, b)
        fin = np.isfinite(b)
        if not np.array_equal(np.isfinite(a), fin):
            return False
        d = np.abs(a - b); d[~fin] = 0
        ok = d <= tol * np.maximum(1, np.abs(np.where(fin, b, 0)))
        if float_lse:
            steps[0] += int((~ok).sum()); steps[1] += int(ok.size)
            for i in np.nonzero(~ok)[0]:
                stepped.append((kind, i if sample is None else sample, int(i)))
            ok = ok | (d <= nstep * FLOAT_STEP)
        return bool(np.all(ok))
    for k in ("log_phased_post", "log_unphased_post", "hap_log_phased_post"):
        assert close(got[k], want[k]), "%s %s" % (what, k)
    assert close(got["hap_log_unphased_post"], want["hap_log_unphased_post"], True, kind="hap"), "%s hap_log_unphased_post" % what
    assert close(got["gl_diff"], want["gl_diff"], True, 2, kind="gldiff"), "%s gl_diff" % what          # a difference of two GLs
    for s in range(len(want["gls"])):
        assert close(got["gls"][s], want["gls"][s], True, kind="gl", sample=s), "%s gls of sample %d" % (what, s)
        assert close(got["phased_gls"][s], want["phased_gls"][s]), "%s phased gls of sample %d" % (what, s)
        gp, wp = np.asarray(got["pls"][s]), np.asarray(want["pls"][s])
        if tol == 0:
            assert np.array_equal(gp, wp), "%s pls of sample %d" % (what, s)
        else:
            bad = np.nonzero(gp != wp)[0]
            g = np.asarray(want["gls"][s]); x = -10 * (g - g.max())
            assert np.all(np.abs(gp[bad] - wp[bad]) <= 1) and np.all(np.abs(x[bad] - np.round(x[bad])) < 10 * FLOAT_STEP), "%s pls of sample %d" % (what, s)
    assert steps[0] <= max(1, 0.005 * steps[1]), "%s: %d of %d values a float step away from the reference" % (what, steps[0], steps[1])
    if verify is not None and stepped:
        # every step must be owed: recompute the reference's cast argument from the oracle's posteriors
        from hipstr_amd import capi
        ora, pb, n_variants, h2a_all = verify
        post, _, map_gt, _ = capi.run_posteriors(ora, "oracle_", pb)
        A_l = np.asarray(pb.a["n_alleles"]); S_l = np.asarray(pb.a["n_samples"]); hap = pb.a["haploid"] if pb.a["haploid"] is not None else np.zeros(len(A_l), np.uint8)
        samp_locus = np.repeat(np.arange(len(A_l)), S_l)
        h2a_off = np.concatenate([[0], np.cumsum(A_l)])
        gl_stepped_samples = set()
        def sample_ctx(s):
            l = int(samp_locus[s]); A = int(A_l[l]); V = int(np.asarray(n_variants)[l])
            P = post[int(pb.post_off[l]) + (s - int(pb.samp_off[l])) * A * A:][:A * A]
            return l, A, V, P, np.asarray(h2a_all)[h2a_off[l]:h2a_off[l] + A]
        for kind, s, i in stepped:
            if kind == "gl":
                l, A, V, P, h2a = sample_ctx(s)
                assert not hap[l], "%s: a haploid GL (equal arguments: no rounding involved) differs, sample %d" % (what, s)
                T = _genotype_totals(P, A, V, h2a)
                i1 = int((np.sqrt(8 * i + 1) - 1) // 2); i2 = i - i1 * (i1 + 1) // 2
                delta = min(T[i1, i2], T[i2, i1]) - max(T[i1, i2], T[i2, i1])
                scale = max(1.0, abs(T[i1, i2]), abs(T[i2, i1]))
                assert float_boundary_distance(delta) <= BOUNDARY_REL * scale, \
                    "%s: GL %d of sample %d differs by %.3g although the reference's cast argument %.17g is %.3g away from any float rounding boundary (arguments of magnitude %.3g)" \
                    % (what, i, s, abs(got["gls"][s][i] - want["gls"][s][i]), delta, float_boundary_distance(delta), scale)
                gl_stepped_samples.add(s)
        for kind, s, i in stepped:
            if kind == "hap":
                l, A, V, P, h2a = sample_ctx(s)
                a, b = int(map_gt[s][0]), int(map_gt[s][1])
                pab, pba = P[a * A + b], P[b * A + a]
                delta = min(pab, pba) - max(pab, pba)
                assert a != b and float_boundary_distance(delta) <= BOUNDARY_REL * max(1.0, abs(pab), abs(pba)), "%s: hap_log_unphased_post of sample %d differs without a rounding boundary (argument %.17g)" % (what, s, delta)
            elif kind == "gldiff":
                assert s in gl_stepped_samples, "%s: GLDIFF of sample %d differs although none of its GLs made an owed float step" % (what, s)
    FLOAT_STEPS_SEEN[0] += steps[0]; FLOAT_STEPS_SEEN[1] += steps[1]
    if steps[0]:            # reported in the pytest summary: how many values needed the widened window
        import warnings
        warnings.warn("%s: %d of %d GL / GLDIFF / unphased-posterior values one float step (<= %g) from the reference%s "
                      "(device exp/log; they vanish with HIPSTR_DEBUG_HOST_LIBM=1: test_float_steps_vanish_with_host_libm)"
                      % (what or "genotype calls", steps[0], steps[1], FLOAT_STEP, ", each verified to sit on a float rounding boundary of the reference's cast" if verify is not None else ""))
    return steps


# ---------------------------------------------------------------- round 5: the device evaluates exp / log correctly rounded (cr_math.h)
def genotypes_identical(got, want):
    """Every output of a genotype-call run equal, bit for bit (NaN == NaN)."""
    eq = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b), equal_nan=np.asarray(b).dtype.kind == "f")
    for k in ("best_hap", "best_gt", "log_phased_post", "log_unphased_post", "hap_log_phased_post", "hap_log_unphased_post", "gl_diff"):
        if not eq(got[k], want[k]):
            return False
    for k in ("gls", "pls", "phased_gls"):
        if len(got[k]) != len(want[k]) or not all(eq(a, b) for a, b in zip(got[k], want[k])):
            return False
    return True


LIBM_NOT_CR = [0, 0]      # over the session: comparisons that needed the second level, comparisons


def assert_genotypes_exact(got, want, run_oracle_cr, what="", verify=None):
    """The contract since round 5 (cr_math.h).  Level 1: the device's genotype calls equal the reference's (golden fixture or the oracle
    with the host libm) bit for bit — tolerance 0.  Where they do not, level 2 must explain it completely: the device equals, bit for bit,
    the oracle run with the SAME correctly rounded exp / log (run_oracle_cr(): an operation-for-operation CPU restatement of the device path),
    and that run differs from the host-libm reference only the way a last-bit difference of an exp / log result can (the host's libm is
    not correctly rounded on ~8 in 10^4 exp arguments: tests/test_cr_math.py) — within 1e-9, a float step only where the reference's
    cast sits on a rounding boundary (assert_genotypes_close with verify).  Returns True if level 1 held."""
    LIBM_NOT_CR[1] += 1
    if genotypes_identical(got, want):
        return True
    LIBM_NOT_CR[0] += 1
    want_cr = run_oracle_cr()
    assert_genotypes_close(got, want_cr, 0, what + " (device vs the oracle with correctly rounded exp/log)")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert_genotypes_close(want_cr, want, 1e-9, what + " (correctly rounded vs host libm)", verify=verify)
    warnings.warn("%s: the host libm is not correctly rounded somewhere in this case — the device equals the oracle evaluated with correctly rounded exp/log bit for bit, "
                  "and that differs from the host-libm reference within the stated tolerance" % (what or "genotype calls"))
    return False


def assert_arrays_exact(got, want, run_oracle_cr, what="", tol=1e-9):
    """Tuples of arrays (posteriors, totals, MAP diplotypes ...) under the same two-level contract as assert_genotypes_exact."""
    same = lambda a, b: np.array_equal(np.asarray(a), np.asarray(b), equal_nan=np.asarray(b).dtype.kind == "f")
    LIBM_NOT_CR[1] += 1
    if all(same(a, b) for a, b in zip(got, want)):
        return True
    LIBM_NOT_CR[0] += 1
    want_cr = run_oracle_cr()
    for i, (a, b) in enumerate(zip(got, want_cr)):
        assert same(a, b), "%s: output %d differs from the oracle with correctly rounded exp/log" % (what, i)
    for i, (a, b) in enumerate(zip(want_cr, want)):
        a = np.asarray(a); b = np.asarray(b)
        if b.dtype.kind != "f":
            assert np.array_equal(a, b), "%s: output %d (correctly rounded vs host libm)" % (what, i)
        else:
            big = b < -1e300
            assert np.array_equal(a < -1e300, big) and np.all(np.abs(a[~big] - b[~big]) <= tol * np.maximum(1, np.abs(b[~big]))), "%s: output %d (correctly rounded vs host libm)" % (what, i)
    import warnings
    warnings.warn("%s: the host libm is not correctly rounded somewhere in this case (device == oracle with correctly rounded exp/log, bit for bit)" % what)
    return False


def batch_with_an_oversized_middle_locus(n_options=1025):
    """Three loci; the middle one has `n_options` STR options (the library takes at most 1024 per block): a CONSISTENT table of an
    unsupported locus — check_locus' refusal, per locus, not validate_tables' (ADVICE r05)."""
    lf = "ACGTTGCATGCATGACCTGAGTCCATGACTTGACA"; rf = "TTGACCGTAGGCTAGGCTTAACGGATCCGATTAGC"
    b = capi.Batch()
    strs = ["AGAT" * 6, "AGAT" * 7, "AGAT" * 5]
    hap = lf + strs[0] + rf
    simple_locus(lf, strs, rf, 4, [(hap[3:83], None, 3, True), (hap[0:80], None, 0, True)], batch=b)
    many = ["AGAT" * 6] + ["AGAT" * (2 + i % 40) + "AG" * (i // 40) for i in range(n_options - 1)]
    simple_locus(lf, many, rf, 4, [(hap[5:85], None, 5, True)], start=900, batch=b)
    simple_locus(lf, strs[:2], rf, 4, [(hap[6:86], None, 6, True), (hap[1:81], None, 1, True), (hap[9:89], None, 9, True)], start=1300, batch=b)
    return b.finalize()
