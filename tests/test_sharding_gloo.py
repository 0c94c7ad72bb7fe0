"""CPU, world_size 2 over gloo: the N>1 path (contiguous cost-balanced locus shards + ordered host-side gather)
reproduces the single-process result exactly.  The oracle stands in for the device here (tests may use it)."""
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from hipstr_amd import capi, shard
from util import synth_to_batch


def test_split_is_contiguous_and_balanced():
    costs = np.array([5, 1, 1, 1, 4, 4, 2, 2], float)
    b = shard.split_loci(costs, 4)
    assert b[0] == 0 and b[-1] == 8 and all(x <= y for x, y in zip(b, b[1:]))
    assert max(costs[lo:hi].sum() for lo, hi in zip(b, b[1:])) <= 8
    assert shard.split_loci(np.ones(3), 8)[-1] == 3      # more ranks than loci: empty tail shards


def test_subset_roundtrip(oracle):
    full = synth_to_batch(capi.SynthBatch(n_loci=5, reads_per_locus=6, n_str_alleles=4, n_flank_opts=2, seed=31, mask_rate=0.2))
    want, wseeds = capi.run_align(oracle, "oracle_", full.ptr, fill=-2.0)
    got, seeds = [], []
    for lo, hi in ((0, 2), (2, 2), (2, 5)):
        sub = shard.batch_from_arrays(shard.subset_arrays(full.arrays, lo, hi))
        if hi > lo:
            p, s = capi.run_align(oracle, "oracle_", sub.ptr, fill=-2.0)
            got.append(p); seeds.append(s)
    assert np.array_equal(np.concatenate(got), want) and np.array_equal(np.concatenate(seeds), wseeds)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = capi.load_oracle()
    full = synth_to_batch(capi.SynthBatch(n_loci=7, reads_per_locus=8, n_str_alleles=5, seed=77))
    res = shard.run_sharded(full.arrays, lambda ptr: capi.run_align(ora, "oracle_", ptr, fill=-2.0), rank, world)
    dist.barrier()
    if rank == 0:
        q.put((res[0], res[1]))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    probs, seeds = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = synth_to_batch(capi.SynthBatch(n_loci=7, reads_per_locus=8, n_str_alleles=5, seed=77))
    want, wseeds = capi.run_align(oracle, "oracle_", full.ptr, fill=-2.0)
    assert np.array_equal(probs, want) and np.array_equal(seeds, wseeds)


def _stage_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hipstr_amd import gen
    import util
    ora = capi.load_oracle()
    em = shard.run_sharded_em(gen.em_case(3, n_loci=7), lambda **kw: capi.run_em(ora, "oracle_", **kw), rank, world)
    nw = shard.run_sharded_nw(gen.nw_pairs(4, n=21), lambda p: capi.run_nw(ora, "oracle_", p, False), rank, world)
    # traceback requests on a one-locus-per-request-capable oracle: shards are cut so that every rank holds whole loci
    full = synth_to_batch(capi.SynthBatch(n_loci=4, reads_per_locus=6, n_str_alleles=3, seed=55))
    _, seeds = capi.run_align(ora, "oracle_", full.ptr)
    rr = [r for r in range(24) if seeds[r] >= 0]; aa = [r % 3 for r in rr]

    def trace_fn(ptr, reads, alleles, h2r):            # the oracle traces one locus per call
        b = ptr._obj if hasattr(ptr, "_obj") else ptr.contents
        ro = np.ctypeslib.as_array(b.read_off, shape=(b.n_loci + 1,))
        sub_arrays = shard_arrays[0]
        out = []
        for r, k in zip(reads, alleles):
            l = int(np.searchsorted(ro, r, side="right") - 1)
            one = shard.batch_from_arrays(shard.subset_arrays(sub_arrays, l, l + 1))
            out += capi.run_trace(ora, "oracle_", one.ptr, [r - int(ro[l])], [k], None)
        return out
    bounds = shard.split_loci(shard.locus_costs(full.arrays), world)
    shard_arrays = [shard.subset_arrays(full.arrays, bounds[rank], bounds[rank + 1])]
    tr = shard.run_sharded_trace(full.arrays, rr, aa, None, trace_fn, rank, world)
    dist.barrier()
    if rank == 0:
        q.put((em, nw, [t["hap_aln"] for t in tr], rr, aa))
    dist.destroy_process_group()


def test_two_rank_gloo_other_stages(oracle):
    """EM loci, Needleman-Wunsch pairs and traceback requests split over 2 ranks give the single-process results in order."""
    from hipstr_amd import gen
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_stage_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    em, nw, tr, rr, aa = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_em = capi.run_em(oracle, "oracle_", **gen.em_case(3, n_loci=7))
    assert all(np.array_equal(a, b) for a, b in zip(em, want_em))
    assert nw == capi.run_nw(oracle, "oracle_", gen.nw_pairs(4, n=21), False)
    full = synth_to_batch(capi.SynthBatch(n_loci=4, reads_per_locus=6, n_str_alleles=3, seed=55))
    want = []
    for r, k in zip(rr, aa):
        l = r // 6
        one = shard.batch_from_arrays(shard.subset_arrays(full.arrays, l, l + 1))
        want.append(capi.run_trace(oracle, "oracle_", one.ptr, [r - 6 * l], [k], None)[0]["hap_aln"])
    assert tr == want
