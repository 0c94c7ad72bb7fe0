"""GPU: the block caches behind every upload (api.hip BlockCache): the cap on idle bytes is enforced (HIPSTR_DEV_CACHE_GIB),
and a driver that refuses a new chunk is survived — a larger free block is reused whatever its size, then idle chunks are given back
and the request tried again — instead of failing while the free list holds what is needed.  Each case in a process of its own
(the caches and their environment are per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r)
from hipstr_amd import capi
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
def stats():
    o = (C.c_int64*12)(); assert hmm.hipstr_debug_cache_stats(o) == 0; return list(o)
def run(n_loci, reads, check=False):
    sb = capi.SynthBatch(n_loci=n_loci, reads_per_locus=reads, n_str_alleles=8, seed=7 + n_loci)
    try:
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    except RuntimeError:
        print("FAILED at", n_loci, reads, hmm.hipstr_last_error(), stats(), file=sys.stderr); raise
    if check:
        want, ws = capi.run_align(ora, "oracle_", sb.ptr)
        assert np.array_equal(gs, ws) and np.array_equal(got, want)
''' % ROOT


def _run(env_extra, body):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", BODY + body], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_idle_bytes_beyond_the_cap_go_back_to_the_driver():
    out = _run({"HIPSTR_DEV_CACHE_GIB": "0.05"}, r'''
run(4, 50, check=True)
held0 = stats()[0]
run(200, 100)                       # hundreds of MB of workspaces
s = stats()
assert s[3] == int(0.05*2**30), s
assert s[2] == 0, s                 # nothing in use between calls
assert s[1] <= s[3] or s[0] <= 256 << 20, s      # idle bytes within the cap (or only the current chunk, which may still be carved from)
run(4, 50, check=True)              # and the cache still works after trimming
print("ok", held0, s)
''')
    assert "ok" in out


def test_driver_refusal_falls_back_to_free_blocks_then_trims():
    """Batches that DOUBLE: a batch's blocks are all larger than the free blocks of the ones before (a request reuses a free block of at most
    twice its size: none qualifies), so the cache carves chunk after chunk until the (simulated) driver refuses at 1500 MiB with a few
    hundred MB in use: get() must give the idle chunks back and take what it needs (the trim-and-retry path).  Then a batch 16 times
    SMALLER with the chunks full: its requests are served from free blocks far beyond the 2x window (the fallback path) — the round-4 cache
    returned NULL in both situations while the free list held what was needed."""
    out = _run({"HIPSTR_DEBUG_DRIVER_LIMIT_MIB": "1500"}, r'''
for n in [25, 50, 100, 200, 400, 25, 400, 50]:
    run(n, 100, check=(n <= 25))
    s = stats()
    assert s[0] <= 1500 << 20, s    # never more than the "driver" has
    assert s[2] == 0, s
s = stats()
assert s[8] > 0 and s[9] > 0, s     # both ways of surviving a refusal were taken
print("ok", s, hmm.hipstr_debug_driver_allocs())
''')
    assert "ok" in out
