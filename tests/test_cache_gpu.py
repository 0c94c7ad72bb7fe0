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
    """The two ways get() survives a driver that refuses a new chunk (simulated at 1050 MiB), block by block: a request is served from a
    free block far beyond the 2x window; a request nothing free can hold makes the cache give its idle chunks back and take what it
    needs.  The round-4 cache returned NULL in both situations while the free list held hundreds of MB.  Then real batches
    of growing size under the same limit: whatever the layout, the cache never holds more than the driver has."""
    out = _run({"HIPSTR_DEBUG_DRIVER_LIMIT_MIB": "1050"}, r'''
MB = 1 << 20
get = lambda n: hmm.hipstr_debug_cache_get(n); put = lambda p: hmm.hipstr_debug_cache_put(C.c_void_p(p))
a = get(300*MB); b = get(300*MB); assert a and b, hmm.hipstr_last_error()
put(a); put(b)                                   # two free blocks (384 MiB each with their headroom) in chunks of 384 and 512 MiB: 154 MiB left at the driver
c = get(150*MB)                                  # 150 MB: beyond the 2x window of the free blocks, no room in a chunk, driver refuses 512 MiB ...
assert c, hmm.hipstr_last_error()
s = stats(); assert s[8] == 1 and s[9] == 0, s   # ... served from a 300 MB free block all the same
d = get(600*MB)                                  # nothing free holds it; the idle 300 MB chunk goes back and the driver has room again
assert d, hmm.hipstr_last_error()
s = stats(); assert s[8] == 1 and s[9] == 1, s
assert s[0] <= 1050*MB, s
put(c); put(d)
e = get(2000*MB)                                 # more than the driver has, whatever is given back: refused, with the driver's message
assert not e and b"out of memory" in hmm.hipstr_last_error()
run(25, 100, check=True)                         # and the cache serves a real batch afterwards
s = stats(); assert s[0] <= 1050*MB and s[2] == 0, s
print("ok", stats(), hmm.hipstr_debug_driver_allocs())
''')
    assert "ok" in out
