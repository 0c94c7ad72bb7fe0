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
