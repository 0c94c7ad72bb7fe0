"""ctypes view of include/hipstr_hmm.h plus a small numpy batch builder.

This module is plumbing for tests and bench.py: it mirrors the two C structs of
the boundary (hipstr_batch_t, hipstr_post_batch_t), loads the product library
(hipstr_amd/csrc/libhipstr_hmm.so — HIP, no CPU fallback) and the bench/test
support libraries.  Loading anything under oracle/ is restricted to tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg (see load_oracle /
load_ref docstrings).
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HMM_LIB = os.environ.get("HIPSTR_HMM_LIB") or os.path.join(ROOT, "hipstr_amd", "csrc", "libhipstr_hmm.so")     # the override: kernel ablation builds (tools/ablate_str.sh)
SYNTH_LIB = os.path.join(ROOT, "hipstr_amd", "synth", "libhipstr_synth.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libhipstr_oracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libhipstr_ref.so")

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class HipstrBatch(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int32),
        ("blk_start", _i32p), ("blk_end", _i32p), ("blk_nopts", _i32p), ("period", _i32p),
        ("stutter", _f64p), ("opt_off", _i32p), ("seq", C.c_char_p), ("hap_off", _i32p), ("realign_hap", _u8p),
        ("read_off", _i32p), ("base_off", _i32p), ("bases", C.c_char_p), ("quals", C.c_char_p),
        ("read_start", _i32p), ("cigar_off", _i32p), ("cigar_op", C.c_char_p), ("cigar_len", _i32p),
        ("realign_read", _u8p),
    ]


class HipstrPostBatch(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int32),
        ("n_alleles", _i32p), ("n_samples", _i32p), ("read_off", _i32p), ("sample_label", _i32p),
        ("log_p1", _f64p), ("log_p2", _f64p), ("read_weight", _i32p), ("log_aln_probs", _f64p), ("haploid", _u8p), ("log_prior", _f64p),
    ]


class HipstrNwBatch(C.Structure):
    _fields_ = [("n_pairs", C.c_int32), ("ref_off", _i32p), ("ref_seqs", C.c_char_p), ("read_off", _i32p), ("read_seqs", C.c_char_p),
                ("use_ref_end_penalty", C.c_int32)]


class HipstrNwOut(C.Structure):
    _fields_ = [("score", C.POINTER(C.c_float)), ("ok", _u8p), ("aln_off", C.POINTER(C.c_int64)), ("ref_al", C.c_char_p), ("read_al", C.c_char_p),
                ("cigar_off", C.POINTER(C.c_int64)), ("cigar_op", C.c_char_p), ("cigar_len", _i32p), ("cap_aln", C.c_int64), ("cap_cigar", C.c_int64)]


def run_nw(lib, prefix, pairs, use_ref_end_penalty=False, unpack=True, timing=None):
    """<prefix>nw_align on [(ref, read), ...] (str) -> list of (score, ok, ref_al, read_al, cigar string)."""
    import time
    n = len(pairs)
    refs = [r.encode() for r, _ in pairs]; reads = [q.encode() for _, q in pairs]
    ro = np.concatenate([[0], np.cumsum([len(r) for r in refs])]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum([len(q) for q in reads])]).astype(np.int32)
    rb = b"".join(refs); qb = b"".join(reads)
    nb = HipstrNwBatch(n, ro.ctypes.data_as(_i32p), rb, qo.ctypes.data_as(_i32p), qb, int(use_ref_end_penalty))
    cap = int(ro[-1] + qo[-1] + 16)
    score = np.zeros(max(n, 1), np.float32); ok = np.zeros(max(n, 1), np.uint8)
    ao = np.zeros(n + 1, np.int64); co = np.zeros(n + 1, np.int64)
    ra = C.create_string_buffer(cap); qa = C.create_string_buffer(cap); cop = C.create_string_buffer(cap); cl = np.zeros(cap, np.int32)
    i64p = C.POINTER(C.c_int64)
    o = HipstrNwOut(score.ctypes.data_as(C.POINTER(C.c_float)), ok.ctypes.data_as(_u8p), ao.ctypes.data_as(i64p), C.cast(ra, C.c_char_p),
                    C.cast(qa, C.c_char_p), co.ctypes.data_as(i64p), C.cast(cop, C.c_char_p), cl.ctypes.data_as(_i32p), cap, cap)
    fn = getattr(lib, prefix + "nw_align")
    fn.restype = C.c_int; fn.argtypes = [C.POINTER(HipstrNwBatch), C.POINTER(HipstrNwOut)]
    t0 = time.perf_counter()
    rc = fn(C.byref(nb), C.byref(o))
    if timing is not None:
        timing["call_s"] = timing.get("call_s", 0.0) + time.perf_counter() - t0
    if rc != 0:
        why = lib.hipstr_last_error().decode() if prefix == "hipstr_" else ""
        raise RuntimeError("%snw_align failed rc=%d %s" % (prefix, rc, why))
    if not unpack:
        return None
    out = []
    copr, rar, qar = cop.raw, ra.raw, qa.raw
    for i in range(n):
        cig = "".join("%d%s" % (cl[k], copr[k:k + 1].decode()) for k in range(co[i], co[i + 1]))
        out.append((float(score[i]), bool(ok[i]), rar[ao[i]:ao[i + 1]].decode(), qar[ao[i]:ao[i + 1]].decode(), cig))
    return out


class HipstrEmBatch(C.Structure):
    _fields_ = [("n_loci", C.c_int32), ("period", _i32p), ("haploid", _u8p), ("n_samples", _i32p), ("read_off", _i32p), ("sample_label", _i32p),
                ("num_bps", _i32p), ("log_p1", _f64p), ("log_p2", _f64p), ("ref_allele", C.c_int32), ("max_iter", C.c_int32),
                ("min_ll_abs_change", C.c_double), ("min_ll_frac_change", C.c_double)]


def run_em(lib, prefix, period, n_samples, read_off, sample_label, num_bps, log_p1, log_p2, haploid=None, ref_allele=0, max_iter=100,
           min_ll_abs_change=0.01, min_ll_frac_change=0.001):
    """<prefix>em_train on a batch of loci -> (trained[n_loci] bool, stutter[n_loci, 6], n_iter[n_loci], final_ll[n_loci])."""
    i32 = lambda x: np.ascontiguousarray(np.asarray(x, np.int32)); f64 = lambda x: np.ascontiguousarray(np.asarray(x, np.float64))
    a = dict(period=i32(period), n_samples=i32(n_samples), read_off=i32(read_off), sample_label=i32(sample_label), num_bps=i32(num_bps),
             log_p1=f64(log_p1), log_p2=f64(log_p2), haploid=None if haploid is None else np.ascontiguousarray(np.asarray(haploid, np.uint8)))
    nl = len(a["period"])
    eb = HipstrEmBatch(nl, a["period"].ctypes.data_as(_i32p), _ptr(a["haploid"], _u8p), a["n_samples"].ctypes.data_as(_i32p),
                       a["read_off"].ctypes.data_as(_i32p), a["sample_label"].ctypes.data_as(_i32p), a["num_bps"].ctypes.data_as(_i32p),
                       a["log_p1"].ctypes.data_as(_f64p), a["log_p2"].ctypes.data_as(_f64p), ref_allele, max_iter, min_ll_abs_change, min_ll_frac_change)
    trained = np.zeros(max(nl, 1), np.uint8); st = np.zeros(max(6 * nl, 6)); it = np.zeros(max(nl, 1), np.int32); ll = np.zeros(max(nl, 1))
    fn = getattr(lib, prefix + "em_train")
    fn.restype = C.c_int; fn.argtypes = [C.POINTER(HipstrEmBatch), _u8p, _f64p, _i32p, _f64p]
    rc = fn(C.byref(eb), trained.ctypes.data_as(_u8p), st.ctypes.data_as(_f64p), it.ctypes.data_as(_i32p), ll.ctypes.data_as(_f64p))
    if rc != 0:
        why = lib.hipstr_last_error().decode() if prefix == "hipstr_" else ""
        raise RuntimeError("%sem_train failed rc=%d %s" % (prefix, rc, why))
    return trained[:nl].astype(bool), st[:6 * nl].reshape(-1, 6), it[:nl], ll[:nl]


class HipstrGtRequest(C.Structure):
    _fields_ = [("n_variants", _i32p), ("hap_to_allele", _i32p), ("calc_gls", C.c_int32), ("calc_pls", C.c_int32), ("calc_phased_gls", C.c_int32)]


class HipstrGtOut(C.Structure):
    _fields_ = [("best_hap", _i32p), ("best_gt", _i32p), ("log_phased_post", _f64p), ("log_unphased_post", _f64p),
                ("hap_log_phased_post", _f64p), ("hap_log_unphased_post", _f64p), ("gl_diff", _f64p), ("gls", _f64p), ("pls", _i32p),
                ("phased_gls", _f64p)]


def run_gt_extract(lib, prefix, pb, n_variants, hap_to_allele, calc_gls=True, calc_pls=True, calc_phased_gls=True):
    """Genotype calls of a PostBatch: <prefix>gt_extract(pb, request, out) for the oracle / reference probe (which compute
    the posteriors themselves), hipstr_post_upload + launch + hipstr_post_extract for the product.  Returns a dict of arrays;
    gls / pls / phased_gls are lists with one array per sample."""
    nl = pb.struct.n_loci
    S = int(pb.samp_off[-1])
    nv = np.ascontiguousarray(np.asarray(n_variants, np.int32)); h2a = np.ascontiguousarray(np.asarray(hap_to_allele, np.int32))
    rq = HipstrGtRequest(nv.ctypes.data_as(_i32p), h2a.ctypes.data_as(_i32p), int(calc_gls), int(calc_pls), int(calc_phased_gls))
    hap = pb.a["haploid"] if pb.a["haploid"] is not None else np.zeros(nl, np.uint8)
    gl_off = [0]; pgl_off = [0]
    for l in range(nl):
        V = int(nv[l])
        for _ in range(int(pb.a["n_samples"][l])):
            gl_off.append(gl_off[-1] + (V if hap[l] else V * (V + 1) // 2)); pgl_off.append(pgl_off[-1] + (V if hap[l] else V * V))
    k = dict(best_hap=np.zeros(max(2 * S, 2), np.int32), best_gt=np.zeros(max(2 * S, 2), np.int32))
    for nm in ("log_phased_post", "log_unphased_post", "hap_log_phased_post", "hap_log_unphased_post", "gl_diff"):
        k[nm] = np.zeros(max(S, 1))
    k["gls"] = np.zeros(max(gl_off[-1], 1)); k["pls"] = np.zeros(max(gl_off[-1], 1), np.int32); k["phased_gls"] = np.zeros(max(pgl_off[-1], 1))
    o = HipstrGtOut(*[k[f].ctypes.data_as(t) for f, t in HipstrGtOut._fields_])
    if prefix == "hipstr_":
        lib.hipstr_post_extract.restype = C.c_int; lib.hipstr_post_extract.argtypes = [C.c_void_p, C.POINTER(HipstrGtRequest), C.POINTER(HipstrGtOut)]
        lib.hipstr_gt_offsets.restype = C.c_int
        lib.hipstr_gt_offsets.argtypes = [C.POINTER(HipstrPostBatch), C.POINTER(HipstrGtRequest), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        go = np.zeros(S + 1, np.int64); pgo = np.zeros(S + 1, np.int64)
        i64p = C.POINTER(C.c_int64)
        assert lib.hipstr_gt_offsets(pb.ptr, C.byref(rq), go.ctypes.data_as(i64p), pgo.ctypes.data_as(i64p)) == 0
        assert list(go) == gl_off and list(pgo) == pgl_off
        pd = lib.hipstr_post_upload(pb.ptr, None)
        if not pd:
            raise RuntimeError("hipstr_post_upload failed: " + lib.hipstr_last_error().decode())
        try:
            rc = lib.hipstr_post_launch(pd, None)
            if rc == 0:
                rc = lib.hipstr_post_extract(pd, C.byref(rq), C.byref(o))
        finally:
            lib.hipstr_post_free(pd)
        if rc != 0:
            raise RuntimeError("hipstr_post_extract failed: " + lib.hipstr_last_error().decode())
    else:
        fn = getattr(lib, prefix + "gt_extract")
        fn.restype = C.c_int; fn.argtypes = [C.POINTER(HipstrPostBatch), C.POINTER(HipstrGtRequest), C.POINTER(HipstrGtOut)]
        rc = fn(pb.ptr, C.byref(rq), C.byref(o))
        if rc != 0:
            raise RuntimeError("%sgt_extract failed rc=%d" % (prefix, rc))
    out = dict(best_hap=k["best_hap"][:2 * S].reshape(-1, 2), best_gt=k["best_gt"][:2 * S].reshape(-1, 2))
    for nm in ("log_phased_post", "log_unphased_post", "hap_log_phased_post", "hap_log_unphased_post", "gl_diff"):
        out[nm] = k[nm][:S]
    out["gls"] = [k["gls"][gl_off[s]:gl_off[s + 1]] for s in range(S)]
    out["pls"] = [k["pls"][gl_off[s]:gl_off[s + 1]] for s in range(S)]
    out["phased_gls"] = [k["phased_gls"][pgl_off[s]:pgl_off[s + 1]] for s in range(S)]
    return out


class HipstrTraceOut(C.Structure):
    _fields_ = [("ll", _f64p), ("max_index", _i32p), ("hap_aln_off", _i32p), ("hap_aln", C.c_char_p), ("stutter_size", _i32p),
                ("str_seq_off", _i32p), ("str_seq", C.c_char_p), ("flank_seq_off", _i32p), ("flank_seq", C.c_char_p),
                ("flank_ins", _i32p), ("flank_del", _i32p), ("indel_off", _i32p), ("indel_pos", _i32p), ("indel_size", _i32p),
                ("snp_off", _i32p), ("snp_pos", _i32p), ("snp_base", C.c_char_p), ("aln_start", _i32p), ("aln_stop", _i32p),
                ("cigar_off", _i32p), ("cigar_op", C.c_char_p), ("cigar_len", _i32p), ("aln_str_off", _i32p), ("aln_str", C.c_char_p),
                ("cap_chars", C.c_int32)]


def run_trace(lib, prefix, bptr, req_read, req_allele, hap_to_ref=None, cap=1 << 16, timing=None, unpack=True, req_seed=None):
    """Call <prefix>trace on a one-locus batch; returns a list of dicts (one per request) with python-typed fields.
    hap_to_ref: list of bytes (one per allele) or None.  The reference probe (prefix 'ref_') always stitches."""
    n = len(req_read)
    o = HipstrTraceOut(); keep = {}
    def i32(name, m):
        a = np.zeros(m, np.int32); keep[name] = a; setattr(o, name, a.ctypes.data_as(_i32p))
    def chars(name):
        a = C.create_string_buffer(cap); keep[name] = a; setattr(o, name, C.cast(a, C.c_char_p))
    keep["ll"] = np.zeros(max(n, 1)); o.ll = keep["ll"].ctypes.data_as(_f64p)
    for nm, m in (("max_index", n), ("hap_aln_off", n + 1), ("stutter_size", n), ("str_seq_off", n + 1), ("flank_seq_off", 2 * n + 1),
                  ("flank_ins", n), ("flank_del", n), ("indel_off", n + 1), ("indel_pos", cap), ("indel_size", cap), ("snp_off", n + 1),
                  ("snp_pos", cap), ("aln_start", n), ("aln_stop", n), ("cigar_off", n + 1), ("cigar_len", cap), ("aln_str_off", n + 1)):
        i32(nm, max(m, 1))
    for nm in ("hap_aln", "str_seq", "flank_seq", "snp_base", "cigar_op", "aln_str"):
        chars(nm)
    o.cap_chars = cap
    rr = np.ascontiguousarray(np.asarray(req_read, np.int32)); aa = np.ascontiguousarray(np.asarray(req_allele, np.int32))
    fn = getattr(lib, prefix + ("trace" if req_seed is None else "trace_seeded"))
    import time
    h2r = None
    if hap_to_ref is not None and prefix != "ref_":
        h2r = (C.c_char_p * len(hap_to_ref))(*hap_to_ref)       # (marshalling of this wrapper, not the call: outside the timed part)
    t_call = time.perf_counter()
    extra, extra_t = [], []
    if req_seed is not None:        # trace_optimal_aln's seed_base argument (HapAligner.h:93)
        ss = np.ascontiguousarray(np.asarray(req_seed, np.int32)); keep["req_seed"] = ss
        extra, extra_t = [ss.ctypes.data_as(_i32p)], [_i32p]
    if prefix == "ref_":
        fn.restype = C.c_int; fn.argtypes = [_BP, C.c_int32, _i32p, _i32p] + extra_t + [C.POINTER(HipstrTraceOut)]
        rc = fn(bptr, n, rr.ctypes.data_as(_i32p), aa.ctypes.data_as(_i32p), *(extra + [C.byref(o)]))
    else:
        fn.restype = C.c_int; fn.argtypes = [_BP, C.c_int32, _i32p, _i32p] + extra_t + [C.POINTER(C.c_char_p), C.POINTER(HipstrTraceOut)]
        rc = fn(bptr, n, rr.ctypes.data_as(_i32p), aa.ctypes.data_as(_i32p), *(extra + [h2r, C.byref(o)]))
    if timing is not None:
        timing["call_s"] = timing.get("call_s", 0.0) + time.perf_counter() - t_call
    if rc != 0:
        why = ""
        if prefix == "hipstr_hmm_":
            lib.hipstr_last_error.restype = C.c_char_p
            why = ": " + lib.hipstr_last_error().decode()
        raise RuntimeError("%strace failed rc=%d%s" % (prefix, rc, why))
    if not unpack:
        return keep
    raws = {nm: keep[nm].raw for nm in ("hap_aln", "str_seq", "flank_seq", "snp_base", "cigar_op", "aln_str")}     # .raw copies: once per pool
    def piece(pool, off, i):
        return raws[pool][keep[off][i]:keep[off][i + 1]].decode()
    out = []
    for q in range(n):
        out.append(dict(
            ll=float(keep["ll"][q]), max_index=int(keep["max_index"][q]), hap_aln=piece("hap_aln", "hap_aln_off", q),
            stutter_size=int(keep["stutter_size"][q]), str_seq=piece("str_seq", "str_seq_off", q),
            flank_left=piece("flank_seq", "flank_seq_off", 2 * q), flank_right=piece("flank_seq", "flank_seq_off", 2 * q + 1),
            flank_ins=int(keep["flank_ins"][q]), flank_del=int(keep["flank_del"][q]),
            indels=[(int(keep["indel_pos"][i]), int(keep["indel_size"][i])) for i in range(keep["indel_off"][q], keep["indel_off"][q + 1])],
            snps=[(int(keep["snp_pos"][i]), raws["snp_base"][i:i + 1].decode()) for i in range(keep["snp_off"][q], keep["snp_off"][q + 1])],
            aln_start=int(keep["aln_start"][q]), aln_stop=int(keep["aln_stop"][q]),
            cigar="".join("%d%s" % (keep["cigar_len"][i], raws["cigar_op"][i:i + 1].decode()) for i in range(keep["cigar_off"][q], keep["cigar_off"][q + 1])),
            aln_str=piece("aln_str", "aln_str_off", q)))
    return out


def ref_hap_aln_info(ref, bptr, n_alleles, cap=1 << 20):
    """Haplotype::get_aln_info() of every allele (list of bytes) from the compiled reference."""
    ref.ref_hap_aln_info.restype = C.c_int
    ref.ref_hap_aln_info.argtypes = [_BP, C.c_char_p, C.c_int, _i32p]
    buf = C.create_string_buffer(cap); offs = np.zeros(n_alleles + 1, np.int32)
    rc = ref.ref_hap_aln_info(bptr, buf, cap, offs.ctypes.data_as(_i32p))
    if rc != 0:
        raise RuntimeError("ref_hap_aln_info rc=%d" % rc)
    raw = buf.raw
    return [raw[offs[k]:offs[k + 1] - 1] for k in range(n_alleles)]


def hap_aln_info(lib, prefix, bptr, cap=1 << 22):
    """<prefix>hap_aln_info: Haplotype::get_aln_info() of every haplotype of every locus of a batch (list of bytes)."""
    b = bptr.contents if hasattr(bptr, "contents") else (bptr._obj if hasattr(bptr, "_obj") else bptr)
    n = int(np.ctypeslib.as_array(b.hap_off, shape=(b.n_loci + 1,))[-1])
    fn = getattr(lib, prefix + "hap_aln_info")
    fn.restype = C.c_int; fn.argtypes = [_BP, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
    buf = C.create_string_buffer(cap); offs = np.zeros(n + 1, np.int64)
    rc = fn(bptr, buf, cap, offs.ctypes.data_as(C.POINTER(C.c_int64)))
    if rc != 0:
        why = ""
        if prefix == "hipstr_":
            lib.hipstr_last_error.restype = C.c_char_p
            why = ": " + lib.hipstr_last_error().decode()
        raise RuntimeError("%shap_aln_info failed rc=%d%s" % (prefix, rc, why))
    raw = buf.raw[:int(offs[n])]          # .raw copies the whole buffer: take it once
    return [raw[offs[k]:offs[k + 1] - 1] for k in range(n)]


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def gray_num_combs(nopts):
    return int(nopts[0]) * int(nopts[1]) * int(nopts[2])


class Batch:
    """Host-side flat batch: numpy arrays + the ctypes struct that points into them."""

    def __init__(self):
        self.blk_start, self.blk_end, self.blk_nopts, self.period, self.stutter = [], [], [], [], []
        self.opt_off, self.seq = [0], bytearray()
        self.hap_off, self.realign_hap = [0], []
        self.read_off, self.base_off, self.bases, self.quals = [0], [0], bytearray(), bytearray()
        self.read_start, self.cigar_off, self.cigar_op, self.cigar_len, self.realign_read = [], [0], bytearray(), [], []
        self._use_hap_mask = False
        self._use_read_mask = False
        self.struct = None

    def add_locus(self, blocks, period, stutter, reads, realign_hap=None):
        """blocks: 3 tuples (start, end, [option sequences]); reads: list of dicts
        {seq, qual, start, cigar:[(op,len)...], realign:bool}."""
        assert len(blocks) == 3
        nopts = []
        for (start, end, opts) in blocks:
            self.blk_start.append(start); self.blk_end.append(end); self.blk_nopts.append(len(opts))
            nopts.append(len(opts))
            for o in opts:
                self.seq += o.encode()
                self.opt_off.append(len(self.seq))
        A = gray_num_combs(nopts)
        self.period.append(period)
        self.stutter += list(stutter)
        self.hap_off.append(self.hap_off[-1] + A)
        if realign_hap is None:
            self.realign_hap += [1] * A
        else:
            assert len(realign_hap) == A
            self._use_hap_mask = True
            self.realign_hap += [1 if x else 0 for x in realign_hap]
        for rd in reads:
            assert len(rd["seq"]) == len(rd["qual"])
            self.bases += rd["seq"].encode(); self.quals += rd["qual"].encode()
            self.base_off.append(len(self.bases))
            self.read_start.append(rd["start"])
            for (op, n) in rd["cigar"]:
                self.cigar_op += op.encode(); self.cigar_len.append(n)
            self.cigar_off.append(len(self.cigar_op))
            flag = rd.get("realign", True)
            if not flag:
                self._use_read_mask = True
            self.realign_read.append(1 if flag else 0)
        self.read_off.append(self.read_off[-1] + len(reads))
        return A

    def finalize(self):
        i32 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.int32))
        a = self.arrays = dict(
            blk_start=i32(self.blk_start), blk_end=i32(self.blk_end), blk_nopts=i32(self.blk_nopts), period=i32(self.period),
            stutter=np.ascontiguousarray(np.array(self.stutter, dtype=np.float64)), opt_off=i32(self.opt_off),
            seq=bytes(self.seq) + b"\0", hap_off=i32(self.hap_off),
            realign_hap=np.array(self.realign_hap, dtype=np.uint8) if self._use_hap_mask else None,
            read_off=i32(self.read_off), base_off=i32(self.base_off), bases=bytes(self.bases) + b"\0", quals=bytes(self.quals) + b"\0",
            read_start=i32(self.read_start), cigar_off=i32(self.cigar_off), cigar_op=bytes(self.cigar_op) + b"\0",
            cigar_len=i32(self.cigar_len if self.cigar_len else [0]),
            realign_read=np.array(self.realign_read, dtype=np.uint8) if self._use_read_mask else None,
        )
        s = HipstrBatch()
        s.n_loci = len(self.period)
        for name in ("blk_start", "blk_end", "blk_nopts", "period", "opt_off", "hap_off", "read_off", "base_off", "read_start",
                     "cigar_off", "cigar_len"):
            setattr(s, name, _ptr(a[name], _i32p))
        s.stutter = _ptr(a["stutter"], _f64p)
        s.seq, s.bases, s.quals, s.cigar_op = a["seq"], a["bases"], a["quals"], a["cigar_op"]
        s.realign_hap = _ptr(a["realign_hap"], _u8p)
        s.realign_read = _ptr(a["realign_read"], _u8p)
        s._keepalive = a     # byref(struct) keeps the struct alive; the struct keeps the arrays alive
        self.struct = s
        return self

    @property
    def ptr(self):
        return C.byref(self.struct)


def batch_dims(bptr):
    """(n_reads, n_out, out_off[n_loci+1]) of a hipstr_batch_t given as ctypes pointer/byref/struct."""
    b = bptr.contents if hasattr(bptr, "contents") else (bptr._obj if hasattr(bptr, "_obj") else bptr)
    n = b.n_loci
    read_off = np.ctypeslib.as_array(b.read_off, shape=(n + 1,))
    hap_off = np.ctypeslib.as_array(b.hap_off, shape=(n + 1,))
    P = np.diff(read_off).astype(np.int64)
    A = np.diff(hap_off).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum(P * A)])
    return int(read_off[-1]), int(out_off[-1]), out_off


class PostBatch:
    def __init__(self, n_alleles, n_samples, read_off, sample_label, log_p1, log_p2, read_weight, log_aln_probs, haploid=None, log_prior=None):
        i32 = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.int32))
        f64 = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        self.a = dict(n_alleles=i32(n_alleles), n_samples=i32(n_samples), read_off=i32(read_off), sample_label=i32(sample_label),
                      log_p1=f64(log_p1), log_p2=f64(log_p2), read_weight=i32(read_weight),
                      log_aln_probs=None if log_aln_probs is None else f64(log_aln_probs),
                      haploid=None if haploid is None else np.ascontiguousarray(np.asarray(haploid, dtype=np.uint8)),
                      log_prior=None if log_prior is None else f64(log_prior))
        s = HipstrPostBatch()
        s.n_loci = len(self.a["n_alleles"])
        for name in ("n_alleles", "n_samples", "read_off", "sample_label", "read_weight"):
            setattr(s, name, _ptr(self.a[name], _i32p))
        s.log_p1 = _ptr(self.a["log_p1"], _f64p); s.log_p2 = _ptr(self.a["log_p2"], _f64p)
        s.log_aln_probs = _ptr(self.a["log_aln_probs"], _f64p)
        s.haploid = _ptr(self.a["haploid"], _u8p)
        s.log_prior = _ptr(self.a["log_prior"], _f64p)
        s._keepalive = self.a
        self.struct = s
        A = self.a["n_alleles"].astype(np.int64); S = self.a["n_samples"].astype(np.int64)
        self.post_off = np.concatenate([[0], np.cumsum(S * A * A)])
        self.samp_off = np.concatenate([[0], np.cumsum(S)])

    @property
    def ptr(self):
        return C.byref(self.struct)


# ----------------------------------------------------------------------------- loaders
def _sig(fn, restype, argtypes):
    fn.restype = restype
    fn.argtypes = argtypes


_BP = C.POINTER(HipstrBatch)
_PBP = C.POINTER(HipstrPostBatch)


def _common_align_sigs(lib, prefix):
    _sig(getattr(lib, prefix + "process_reads"), C.c_int, [_BP, _f64p, _i32p])
    _sig(getattr(lib, prefix + "posteriors"), C.c_int, [_PBP, _f64p, _f64p, _i32p, _f64p])


_scalar_probes = {
    "int_log": (C.c_double, [C.c_int]),
    "transition": (C.c_double, [C.c_int, C.c_int]),
    "base_quality": (C.c_double, [C.c_int, C.c_int]),
    "stutter_pmf": (C.c_double, [_f64p, C.c_int, C.c_int, C.c_int]),
    "fast_lse_vec": (C.c_double, [_f64p, C.c_int]),
    "fast_lse2": (C.c_double, [C.c_double, C.c_double]),
    "log_sum_exp": (C.c_double, [_f64p, C.c_int]),
}


def load_oracle():
    """oracle/libhipstr_oracle.so — the C restatement.  TEST INFRASTRUCTURE: callers must be
    tests/, __graft_entry__.smoke() or bench.py's cpu_baseline leg."""
    lib = C.CDLL(ORACLE_LIB)
    _common_align_sigs(lib, "oracle_")
    _sig(lib.oracle_calc_seed_bases, C.c_int, [_BP, _i32p])
    _sig(lib.oracle_allele_options, C.c_int, [_i32p, C.c_int, _i32p])
    _sig(lib.oracle_debug_row_h, C.c_int, [_BP, C.c_int, _i32p, _i32p, C.c_int])
    for name, (rt, at) in _scalar_probes.items():
        _sig(getattr(lib, "oracle_" + name), rt, at)
    _sig(lib.oracle_set_cr_math, None, [C.c_int])
    return lib


class oracle_cr_math:
    """with capi.oracle_cr_math(oracle): ... — the oracle's posterior / genotype-call / EM stages evaluate exp and log with the correctly
    rounded functions of hipstr_amd/csrc/cr_math.h (what the device kernels use) instead of the host libm: an operation-for-operation CPU
    restatement of the device path.  Only for those stages: the alignment path's model tables stay the host libm's in the library too."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        self.lib.oracle_set_cr_math(1)
        return self.lib

    def __exit__(self, *a):
        self.lib.oracle_set_cr_math(0)
        return False


def load_ref():
    """oracle/_ref/libhipstr_ref.so — the real reference sources compiled by oracle/Makefile.
    genotyper.cpp leaves FastaReader/htslib symbols undefined (only get_vcf_header uses them), so
    the library is opened with lazy binding.  TEST INFRASTRUCTURE, same rule as load_oracle."""
    libc = C.CDLL(None)
    libc.dlopen.restype = C.c_void_p
    libc.dlopen.argtypes = [C.c_char_p, C.c_int]
    handle = libc.dlopen(REF_LIB.encode(), os.RTLD_LAZY)
    if not handle:
        raise OSError("cannot dlopen " + REF_LIB)
    lib = C.CDLL(REF_LIB, handle=handle)
    _common_align_sigs(lib, "ref_")
    _sig(lib.ref_hap_sequences, C.c_int, [_BP, C.c_int, C.c_char_p, C.c_int, _i32p])
    _sig(lib.ref_log_thresh, C.c_double, [])
    _sig(lib.ref_log_one_half, C.c_double, [])
    for name, (rt, at) in _scalar_probes.items():
        _sig(getattr(lib, "ref_" + name), rt, at)
    return lib


def have_ref():
    return os.path.exists(REF_LIB)


def load_synth():
    lib = C.CDLL(SYNTH_LIB)
    _sig(lib.synth_create, C.c_void_p, [C.c_int32] * 7 + [C.c_uint64, C.c_double])
    _sig(lib.synth_create_at, C.c_void_p, [C.c_int32] * 8 + [C.c_uint64, C.c_double])
    _sig(lib.synth_batch, _BP, [C.c_void_p])
    _sig(lib.synth_src_allele, _i32p, [C.c_void_p])
    _sig(lib.synth_free, None, [C.c_void_p])
    return lib


class SynthBatch:
    """Seeded synthetic loci (hipstr_amd/synth/synth.cpp).  .ptr is a hipstr_batch_t*."""

    def __init__(self, n_loci, reads_per_locus, n_str_alleles, read_len=150, flank_len=60, str_bp=40, n_flank_opts=1,
                 seed=20260928, mask_rate=0.0, first_locus=0):
        self.lib = load_synth()
        self.h = self.lib.synth_create_at(first_locus, n_loci, reads_per_locus, n_str_alleles, read_len, flank_len, str_bp, n_flank_opts, seed, mask_rate)
        self.ptr = self.lib.synth_batch(self.h)
        self.n_reads, self.n_out, self.out_off = batch_dims(self.ptr)
        self.n_loci = n_loci

    def src_allele(self):
        return np.ctypeslib.as_array(self.lib.synth_src_allele(self.h), shape=(self.n_reads,)).copy()

    def close(self):
        if self.h:
            self.lib.synth_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_hmm():
    """The product: hipstr_amd/csrc/libhipstr_hmm.so (HIP/gfx950).  Raises if it is missing —
    there is deliberately no CPU fallback."""
    if not os.path.exists(HMM_LIB):
        raise RuntimeError("libhipstr_hmm.so is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(HMM_LIB)
    _sig(lib.hipstr_batch_out_offsets, C.c_int, [_BP, C.POINTER(C.c_int64)])
    _sig(lib.hipstr_hmm_init, C.c_int, [C.c_int])
    _sig(lib.hipstr_hmm_shutdown, None, [])
    _sig(lib.hipstr_hmm_trim, C.c_int64, [])
    _sig(lib.hipstr_hmm_upload, C.c_void_p, [_BP])
    _sig(lib.hipstr_hmm_free, None, [C.c_void_p])
    _sig(lib.hipstr_hmm_align, C.c_int, [C.c_void_p, C.c_void_p])
    _sig(lib.hipstr_hmm_align_timed, C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)])
    _sig(lib.hipstr_hmm_fetch, C.c_int, [C.c_void_p, _f64p, _i32p])
    _sig(lib.hipstr_hmm_dev_aln_probs, C.c_void_p, [C.c_void_p])
    _sig(lib.hipstr_hmm_process_reads, C.c_int, [_BP, _f64p, _i32p])
    _sig(lib.hipstr_hmm_process_reads_seeded, C.c_int, [_BP, _i32p, _f64p, _i32p])
    _sig(lib.hipstr_hmm_upload_seeded, C.c_void_p, [_BP, _i32p])
    _sig(lib.hipstr_calc_seed_bases, C.c_int, [_BP, _i32p])
    _sig(lib.hipstr_post_offsets, C.c_int, [_PBP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
    _sig(lib.hipstr_post_run, C.c_int, [_PBP, C.c_void_p, _f64p, _f64p, _i32p, _f64p])
    _sig(lib.hipstr_post_upload, C.c_void_p, [_PBP, C.c_void_p])
    _sig(lib.hipstr_post_launch, C.c_int, [C.c_void_p, C.c_void_p])
    _sig(lib.hipstr_post_fetch, C.c_int, [C.c_void_p, _f64p, _f64p, _i32p, _f64p])
    _sig(lib.hipstr_post_free, None, [C.c_void_p])
    _sig(lib.hipstr_hmm_profile, C.c_int, [C.c_void_p, C.c_int])
    _sig(lib.hipstr_hmm_profile_read, C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int])
    _sig(lib.hipstr_hmm_workload, C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
    _sig(lib.hipstr_debug_rows, C.c_int, [_BP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int])
    _sig(lib.hipstr_debug_prepare, C.c_int, [_BP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)])
    _sig(lib.hipstr_debug_str_groups, C.c_int, [_BP, _i32p, _i32p, _i32p, C.c_int, _i32p, C.c_int, _i32p])
    _sig(lib.hipstr_debug_simple_table, C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)])
    _sig(lib.hipstr_last_error, C.c_char_p, [])
    _sig(lib.hipstr_debug_driver_allocs, C.c_int64, [])
    _sig(lib.hipstr_locus_costs, C.c_int, [_BP, _f64p])
    _sig(lib.hipstr_debug_cr_math, C.c_int, [C.c_int, _f64p, _f64p, C.c_int64])
    _sig(lib.hipstr_debug_cache_get, C.c_void_p, [C.c_int64])
    _sig(lib.hipstr_debug_cache_put, None, [C.c_void_p])
    _sig(lib.hipstr_debug_cache_stats, C.c_int, [C.POINTER(C.c_int64)])
    _sig(lib.hipstr_debug_allele_kinds, C.c_int, [C.c_void_p, C.POINTER(C.c_int64)])
    _sig(lib.hipstr_debug_stream_create, C.c_void_p, [])
    _sig(lib.hipstr_debug_stream_destroy, None, [C.c_void_p])
    _sig(lib.hipstr_debug_fetch_table, C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64])
    return lib


def _why(lib, prefix):
    """': <hipstr_last_error()>' for the MI355X library's entry points."""
    if not prefix.startswith("hipstr_"):
        return ""
    lib.hipstr_last_error.restype = C.c_char_p
    return ": " + lib.hipstr_last_error().decode()


def run_align(lib, prefix, bptr, fill=np.nan, seed_in=None):
    """Call <prefix>process_reads on a batch; returns (aln_probs, seeds) numpy arrays.
    Entries the callee leaves untouched keep `fill`.  seed_in: per-read seed bases chosen by the caller
    (<prefix>process_reads_seeded = HapAligner::process_read's seed_base argument; -2 = compute)."""
    n_reads, n_out, _ = batch_dims(bptr)
    probs = np.full(max(n_out, 1), fill, dtype=np.float64)
    seeds = np.full(max(n_reads, 1), -7, dtype=np.int32)
    if seed_in is not None:
        si = np.ascontiguousarray(np.asarray(seed_in, np.int32))
        fn = getattr(lib, prefix + "process_reads_seeded")
        fn.restype = C.c_int; fn.argtypes = [_BP, _i32p, _f64p, _i32p]
        rc = fn(bptr, si.ctypes.data_as(_i32p), probs.ctypes.data_as(_f64p), seeds.ctypes.data_as(_i32p))
        if rc != 0:
            raise RuntimeError("%sprocess_reads_seeded failed rc=%d%s" % (prefix, rc, _why(lib, prefix)))
        return probs[:n_out], seeds[:n_reads]
    rc = getattr(lib, prefix + "process_reads")(bptr, probs.ctypes.data_as(_f64p), seeds.ctypes.data_as(_i32p))
    if rc != 0:
        raise RuntimeError("%sprocess_reads failed rc=%d%s" % (prefix, rc, _why(lib, prefix)))
    return probs[:n_out], seeds[:n_reads]


class HipstrStreamOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("slots", C.c_int32), ("batch_alignments", C.c_int64)]


class HipstrStreamStats(C.Structure):
    _fields_ = [("batches", C.c_int64), ("tickets", C.c_int64), ("alignment_slots", C.c_int64), ("host_seconds", C.c_double),
                ("wait_seconds", C.c_double), ("open_seconds", C.c_double), ("cpu_submit_seconds", C.c_double), ("cpu_prepare_seconds", C.c_double),
                ("cpu_upload_seconds", C.c_double), ("cpu_collect_seconds", C.c_double)]


class Stream:
    """hipstr_stream_*: loci in, results out in submission order (include/hipstr_hmm.h)."""

    def __init__(self, lib, device=0, slots=0, batch_alignments=0):
        self.lib = lib
        _sig(lib.hipstr_stream_open, C.c_void_p, [C.POINTER(HipstrStreamOpts)])
        _sig(lib.hipstr_stream_submit, C.c_int64, [C.c_void_p, _BP])
        _sig(lib.hipstr_stream_flush, C.c_int, [C.c_void_p])
        _sig(lib.hipstr_stream_submit_each, C.c_int, [C.c_void_p, _BP, C.POINTER(C.c_int64)])
        _sig(lib.hipstr_stream_collect, C.c_int, [C.c_void_p, C.c_int64, _f64p, C.c_int64, _i32p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
        _sig(lib.hipstr_stream_next_size, C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
        _sig(lib.hipstr_stream_next, C.c_int, [C.c_void_p, C.POINTER(C.c_int64), _f64p, C.c_int64, _i32p, C.c_int64])
        _sig(lib.hipstr_stream_stats, C.c_int, [C.c_void_p, C.POINTER(HipstrStreamStats)])
        _sig(lib.hipstr_stream_close, C.c_int, [C.c_void_p])
        o = HipstrStreamOpts(device, slots, batch_alignments)
        self.h = lib.hipstr_stream_open(C.byref(o))
        if not self.h:
            raise RuntimeError("hipstr_stream_open failed: " + lib.hipstr_last_error().decode())

    def submit(self, bptr):
        t = self.lib.hipstr_stream_submit(self.h, bptr)
        if t < 0:
            raise RuntimeError("hipstr_stream_submit failed: " + self.lib.hipstr_last_error().decode())
        return t

    def flush(self):
        assert self.lib.hipstr_stream_flush(self.h) == 0

    def submit_each(self, bptr):
        """Every locus of the batch as its own submission; returns the first ticket."""
        t = C.c_int64(-1)
        if self.lib.hipstr_stream_submit_each(self.h, bptr, C.byref(t)) != 0:
            raise RuntimeError("hipstr_stream_submit_each failed: " + self.lib.hipstr_last_error().decode())
        return t.value

    def collect(self, n_tickets, probs, seeds):
        """The next n_tickets submissions in order, back to back into probs / seeds; returns (doubles, seeds) written."""
        a = C.c_int64(); b = C.c_int64()
        if self.lib.hipstr_stream_collect(self.h, n_tickets, probs.ctypes.data_as(_f64p), probs.size, seeds.ctypes.data_as(_i32p), seeds.size,
                                          C.byref(a), C.byref(b)) != 0:
            raise RuntimeError("hipstr_stream_collect failed: " + self.lib.hipstr_last_error().decode())
        return a.value, b.value

    def next(self, fill=np.nan, into=None):
        """(ticket, aln_probs, seeds) of the next submission in order, or None when nothing is outstanding."""
        t = C.c_int64(); n_out = C.c_int64(); n_reads = C.c_int64()
        if self.lib.hipstr_stream_next_size(self.h, C.byref(t), C.byref(n_out), C.byref(n_reads)) == 2:
            return None
        if into is None:
            probs = np.full(max(n_out.value, 1), fill); seeds = np.full(max(n_reads.value, 1), -7, np.int32)
        else:
            probs, seeds = into
        rc = self.lib.hipstr_stream_next(self.h, C.byref(t), probs.ctypes.data_as(_f64p), probs.size, seeds.ctypes.data_as(_i32p), seeds.size)
        if rc != 0:
            raise RuntimeError("hipstr_stream_next failed: " + self.lib.hipstr_last_error().decode())
        return t.value, probs[:n_out.value], seeds[:n_reads.value]

    def stats(self):
        st = HipstrStreamStats()
        assert self.lib.hipstr_stream_stats(self.h, C.byref(st)) == 0
        return {k: getattr(st, k) for k, _ in HipstrStreamStats._fields_}

    def close(self):
        if self.h:
            self.lib.hipstr_stream_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_posteriors(lib, prefix, pb):
    S = int(pb.samp_off[-1])
    post = np.zeros(max(int(pb.post_off[-1]), 1)); tot = np.zeros(max(S, 1)); gt = np.zeros(max(2 * S, 2), dtype=np.int32)
    ltot = np.zeros(max(pb.struct.n_loci, 1))
    fn = getattr(lib, prefix + "posteriors")
    rc = fn(pb.ptr, post.ctypes.data_as(_f64p), tot.ctypes.data_as(_f64p), gt.ctypes.data_as(_i32p), ltot.ctypes.data_as(_f64p))
    if rc != 0:
        raise RuntimeError("%sposteriors failed rc=%d" % (prefix, rc))
    return post[:int(pb.post_off[-1])], tot[:S], gt[:2 * S].reshape(-1, 2), ltot[:pb.struct.n_loci]
