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
