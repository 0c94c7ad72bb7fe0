"""GPU: SURVEY §8(f)4 end to end — a shard written by one process and read back by another (hipstr_batch_write / _read) is fed, locus
by locus, to TWO device streams the way two workers of a sharded run would take alternate blocks of the region list; each worker's
results become records pushed into the ordered gather (hipstr_gather_*); what comes out must be the single-stream order and, record by
record, bit-identical to the one-shot call and to the oracle."""
import ctypes as C

import numpy as np
import pytest

from hipstr_amd import capi, shard
import util
from test_batch_io import _api as _io_api
from test_gather import _api as _gather_api, _pop

pytestmark = pytest.mark.gpu
FILL = -5.5


def test_written_shard_through_two_streams_and_the_gather(hmm, oracle, tmp_path):
    lib = _gather_api(_io_api(hmm))
    sb = capi.SynthBatch(n_loci=24, reads_per_locus=18, n_str_alleles=6, n_flank_opts=2, seed=91, mask_rate=0.15)
    path = str(tmp_path / "shard.hsb").encode()
    assert lib.hipstr_batch_write(path, sb.ptr) == 0
    f = lib.hipstr_batch_read(path)
    assert f, lib.hipstr_last_error()
    back = lib.hipstr_batch_file_batch(f)                        # what the GPU-owning process sees
    view = type("V", (), {"ptr": back})()
    arrays = util.synth_to_batch(view).arrays
    pieces = [shard.batch_from_arrays(shard.subset_arrays(arrays, l, l + 1)) for l in range(24)]
    want_oracle = [capi.run_align(oracle, "oracle_", p.ptr, fill=FILL) for p in pieces]
    want_one = [capi.run_align(hmm, "hipstr_hmm_", p.ptr, fill=FILL) for p in pieces]
    # region coordinates of the loci: chromosome 0/1, ascending positions
    coords = [(l // 14, 1000 + 137 * l) for l in range(24)]
    # blocks of 1..4 consecutive regions dealt alternately to two workers, each with its own stream
    rng = np.random.default_rng(3)
    owner = []
    w = 0
    while len(owner) < 24:
        owner += [w % 2] * int(rng.integers(1, 5)); w += 1
    owner = owner[:24]
    streams = [capi.Stream(hmm, slots=2, batch_alignments=600), capi.Stream(hmm, slots=2, batch_alignments=600)]
    for l in range(24):
        streams[owner[l]].submit(pieces[l].ptr)
    g = lib.hipstr_gather_open(2)
    out = []
    def drain():
        while True:
            rc, rec = _pop(lib, g, cap=1 << 16)
            if rc != 0:
                return rc
            out.append(rec)
    for wk in (0, 1):
        mine = [l for l in range(24) if owner[l] == wk]
        for l in mine:                                            # a worker emits its records in its own (ascending) order
            t, probs, seeds = streams[wk].next(fill=FILL)
            payload = np.concatenate([probs.view(np.uint8), seeds.view(np.uint8)]).tobytes()
            assert lib.hipstr_gather_push(g, wk, coords[l][0], coords[l][1], payload, len(payload)) == 0
            drain()
        assert lib.hipstr_gather_end(g, wk) == 0
        streams[wk].close()
    assert drain() == 3                                           # every stream ended, nothing pending
    lib.hipstr_gather_close(g)
    assert [(c, p) for _, c, p, _ in out] == coords               # the single-stream (= region) order
    for l, (_, _, _, payload) in enumerate(out):
        wp, ws = want_one[l]
        probs = np.frombuffer(payload[:8 * wp.size], np.float64); seeds = np.frombuffer(payload[8 * wp.size:], np.int32)
        assert np.array_equal(probs, wp) and np.array_equal(seeds, ws), "locus %d vs the one-shot call" % l
        assert np.array_equal(probs, want_oracle[l][0]) and np.array_equal(seeds, want_oracle[l][1]), "locus %d vs the oracle" % l
    lib.hipstr_batch_file_free(f)
