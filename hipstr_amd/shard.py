"""Sharding of a locus batch across GPUs (one process per GPU) and the host-side ordered gather.

STR loci are independent (the reference's own scale-out is "N processes on N BED shards", README.md:167-171), so
the region list is cut into `world` contiguous, cost-balanced chunks; no collective touches the data path.  The
only communication is the gather of per-rank results on rank 0 in rank order, which — the chunks being
contiguous — is locus order, the order the reference's VCF writer needs (vcf_writer.cpp:7-36)."""
import numpy as np

from . import capi
from .capi import HipstrBatch, _i32p, _f64p, _u8p


def locus_costs(a):
    """Work estimate per locus — the library's own (hipstr_locus_costs, prep.cpp locus_cost): reads x realigned alleles x read length x
    [flank rows + STR block priced by its interruptions], what hipstr_multi_submit deals blocks by as well.  Host-only."""
    hmm = capi.load_hmm()
    b = batch_from_arrays(dict(a))
    out = np.zeros(len(a["period"]), np.float64)
    if hmm.hipstr_locus_costs(b.ptr, out.ctypes.data_as(_f64p)) != 0:
        raise RuntimeError("hipstr_locus_costs: " + hmm.hipstr_last_error().decode())
    return np.maximum(out, 1e-3)


def split_loci(costs, world):
    """Boundaries [b_0=0, ..., b_world=n] of contiguous chunks with near-equal total cost."""
    costs = np.asarray(costs, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(cum, cum[-1] * r / world, side="left")))
    bounds.append(len(costs))
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def subset_arrays(a, lo, hi):
    """Arrays (as in capi.Batch.arrays) of loci [lo, hi) of a batch, re-based to start at zero."""
    out = {}
    out["blk_start"] = a["blk_start"][3 * lo:3 * hi].copy(); out["blk_end"] = a["blk_end"][3 * lo:3 * hi].copy()
    out["blk_nopts"] = a["blk_nopts"][3 * lo:3 * hi].copy(); out["period"] = a["period"][lo:hi].copy()
    out["stutter"] = a["stutter"][6 * lo:6 * hi].copy()
    o0 = int(a["blk_nopts"][:3 * lo].sum()); o1 = o0 + int(out["blk_nopts"].sum())
    s0, s1 = int(a["opt_off"][o0]), int(a["opt_off"][o1])
    out["opt_off"] = (a["opt_off"][o0:o1 + 1] - s0).astype(np.int32)
    out["seq"] = bytes(a["seq"][s0:s1]) + b"\0"
    h0, h1 = int(a["hap_off"][lo]), int(a["hap_off"][hi])
    out["hap_off"] = (a["hap_off"][lo:hi + 1] - h0).astype(np.int32)
    out["realign_hap"] = None if a["realign_hap"] is None else a["realign_hap"][h0:h1].copy()
    r0, r1 = int(a["read_off"][lo]), int(a["read_off"][hi])
    out["read_off"] = (a["read_off"][lo:hi + 1] - r0).astype(np.int32)
    b0, b1 = int(a["base_off"][r0]), int(a["base_off"][r1])
    out["base_off"] = (a["base_off"][r0:r1 + 1] - b0).astype(np.int32)
    out["bases"] = bytes(a["bases"][b0:b1]) + b"\0"; out["quals"] = bytes(a["quals"][b0:b1]) + b"\0"
    out["read_start"] = a["read_start"][r0:r1].copy()
    c0, c1 = int(a["cigar_off"][r0]), int(a["cigar_off"][r1])
    out["cigar_off"] = (a["cigar_off"][r0:r1 + 1] - c0).astype(np.int32)
    out["cigar_op"] = bytes(a["cigar_op"][c0:c1]) + b"\0"
    out["cigar_len"] = a["cigar_len"][c0:c1].copy() if c1 > c0 else np.zeros(1, np.int32)
    out["realign_read"] = None if a["realign_read"] is None else a["realign_read"][r0:r1].copy()
    return out


def batch_from_arrays(a):
    b = capi.Batch()
    b.arrays = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in a.items()}
    a = b.arrays
    s = HipstrBatch()
    s.n_loci = len(a["period"])
    for k in ("blk_start", "blk_end", "blk_nopts", "period", "opt_off", "hap_off", "read_off", "base_off", "read_start", "cigar_off", "cigar_len"):
        setattr(s, k, a[k].ctypes.data_as(_i32p))
    s.stutter = a["stutter"].ctypes.data_as(_f64p)
    s.seq, s.bases, s.quals, s.cigar_op = a["seq"], a["bases"], a["quals"], a["cigar_op"]
    s.realign_hap = None if a["realign_hap"] is None else a["realign_hap"].ctypes.data_as(_u8p)
    s.realign_read = None if a["realign_read"] is None else a["realign_read"].ctypes.data_as(_u8p)
    s._keepalive = a
    b.struct = s
    return b


def run_sharded(arrays, align_fn, rank, world, group=None):
    """Every rank aligns its contiguous chunk of loci with align_fn(batch_ptr) -> (aln_probs, seeds); rank 0 returns the
    results of the whole batch in locus order (other ranks return None).  No collective on the data path: one gather."""
    bounds = split_loci(locus_costs(arrays), world)
    lo, hi = bounds[rank], bounds[rank + 1]
    local = batch_from_arrays(subset_arrays(arrays, lo, hi))
    probs, seeds = align_fn(local.ptr) if hi > lo else (np.zeros(0), np.zeros(0, np.int32))
    if world == 1:
        return probs, seeds
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((probs, seeds), gathered, dst=0, group=group)
    if rank != 0:
        return None
    return np.concatenate([g[0] for g in gathered]), np.concatenate([g[1] for g in gathered])


def _gather_in_rank_order(local, rank, world, group):
    if world == 1:
        return [local]
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0, group=group)
    return gathered if rank == 0 else None


def em_subset(kw, lo, hi):
    """Loci [lo, hi) of an EM input (the keyword arguments of capi.run_em), reads re-based."""
    ro = np.asarray(kw["read_off"])
    r0, r1 = int(ro[lo]), int(ro[hi])
    out = dict(kw)
    for k in ("period", "n_samples", "haploid"):
        if kw.get(k) is not None:
            out[k] = np.asarray(kw[k])[lo:hi]
    out["read_off"] = ro[lo:hi + 1] - r0
    for k in ("sample_label", "num_bps", "log_p1", "log_p2"):
        out[k] = np.asarray(kw[k])[r0:r1]
    return out


def run_sharded_em(kw, em_fn, rank, world, group=None):
    """De novo stutter EM over ranks: contiguous chunks of loci balanced by reads x (distinct sizes)^2; em_fn(**kw) ->
    (trained, stutter, n_iter, final_ll).  Rank 0 returns the whole batch in locus order."""
    ro = np.asarray(kw["read_off"]); nb = np.asarray(kw["num_bps"])
    costs = [max(1.0, float(ro[l + 1] - ro[l]) * (len(set(nb[ro[l]:ro[l + 1]].tolist())) + 1) ** 2) for l in range(len(ro) - 1)]
    bounds = split_loci(costs, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    local = em_fn(**em_subset(kw, lo, hi)) if hi > lo else (np.zeros(0, bool), np.zeros((0, 6)), np.zeros(0, np.int32), np.zeros(0))
    parts = _gather_in_rank_order(local, rank, world, group)
    if parts is None:
        return None
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(4))


def run_sharded_nw(pairs, nw_fn, rank, world, group=None):
    """Needleman-Wunsch pairs over ranks, balanced by cells; nw_fn(pairs) -> list of results.  Rank 0 returns them in input order."""
    bounds = split_loci([len(r) * len(q) for r, q in pairs], world)
    local = nw_fn(pairs[bounds[rank]:bounds[rank + 1]]) if bounds[rank + 1] > bounds[rank] else []
    parts = _gather_in_rank_order(local, rank, world, group)
    return None if parts is None else [x for p in parts for x in p]


def run_sharded_trace(arrays, req_read, req_allele, hap_to_ref, trace_fn, rank, world, group=None):
    """Traceback requests over ranks: the locus shards of run_sharded, every rank tracing the requests that name its reads.
    trace_fn(batch_ptr, req_read, req_allele, hap_to_ref) -> list of per-request results.  Rank 0 returns them in request order."""
    bounds = split_loci(locus_costs(arrays), world)
    lo, hi = bounds[rank], bounds[rank + 1]
    r0, r1 = int(arrays["read_off"][lo]), int(arrays["read_off"][hi])
    h0, h1 = int(arrays["hap_off"][lo]), int(arrays["hap_off"][hi])
    mine = [i for i, r in enumerate(req_read) if r0 <= r < r1]
    if mine:
        sub = batch_from_arrays(subset_arrays(arrays, lo, hi))
        res = trace_fn(sub.ptr, [req_read[i] - r0 for i in mine], [req_allele[i] for i in mine], None if hap_to_ref is None else hap_to_ref[h0:h1])
    else:
        res = []
    parts = _gather_in_rank_order((mine, res), rank, world, group)
    if parts is None:
        return None
    out = [None] * len(req_read)
    for idx, res in parts:
        for i, x in zip(idx, res):
            out[i] = x
    return out
