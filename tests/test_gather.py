"""Host only: hipstr_gather_* — k-way merge of per-worker record streams by (chromosome, position), the ordered gather in front of
the VCF writer (vcf_writer.cpp:7-36 tolerates disorder only within 50 bp)."""
import ctypes as C

import numpy as np

from hipstr_amd import capi


def _api(lib):
    lib.hipstr_gather_open.restype = C.c_void_p; lib.hipstr_gather_open.argtypes = [C.c_int32]
    lib.hipstr_gather_push.restype = C.c_int; lib.hipstr_gather_push.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    lib.hipstr_gather_end.restype = C.c_int; lib.hipstr_gather_end.argtypes = [C.c_void_p, C.c_int32]
    lib.hipstr_gather_pop.restype = C.c_int
    lib.hipstr_gather_pop.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.hipstr_gather_close.restype = None; lib.hipstr_gather_close.argtypes = [C.c_void_p]
    return lib


def _pop(lib, g, cap=256):
    s = C.c_int32(); c = C.c_int32(); p = C.c_int32(); n = C.c_int64(); buf = C.create_string_buffer(cap)
    rc = lib.hipstr_gather_pop(g, C.byref(s), C.byref(c), C.byref(p), buf, cap, C.byref(n))
    return rc, (s.value, c.value, p.value, buf.raw[:n.value]) if rc == 0 else None


def test_merge_of_interleaved_shards(hmm_host):
    lib = _api(hmm_host)
    rng = np.random.default_rng(4)
    # a sorted region list dealt to 4 workers in round-robin blocks of 1..5 regions (the load-balanced sharding of SURVEY §8e)
    regions = sorted({(int(c), int(p)) for c, p in zip(rng.integers(0, 3, 300), rng.integers(0, 10 ** 6, 300))})
    shards = [[] for _ in range(4)]
    i = 0; w = 0
    while i < len(regions):
        k = int(rng.integers(1, 6)); shards[w % 4] += regions[i:i + k]; i += k; w += 1
    g = lib.hipstr_gather_open(4)
    cursors = [0] * 4; out = []
    while True:
        rc, rec = _pop(lib, g)
        if rc == 0:
            out.append(rec); continue
        if rc == 3:
            break
        assert rc == 2
        # feed whichever streams are dry, a few records at a time, out of step with each other
        for s in range(4):
            for _ in range(int(rng.integers(1, 4))):
                if cursors[s] < len(shards[s]):
                    c, p = shards[s][cursors[s]]; cursors[s] += 1
                    assert lib.hipstr_gather_push(g, s, c, p, b"%d:%d" % (c, p), len(b"%d:%d" % (c, p))) == 0
            if cursors[s] == len(shards[s]):
                lib.hipstr_gather_end(g, s)
    assert [(c, p) for _, c, p, _ in out] == regions
    assert all(payload == b"%d:%d" % (c, p) for _, c, p, payload in out)
    lib.hipstr_gather_close(g)


def test_order_violations_and_small_buffers(hmm_host):
    lib = _api(hmm_host)
    g = lib.hipstr_gather_open(2)
    assert lib.hipstr_gather_push(g, 0, 1, 500, b"abc", 3) == 0
    assert lib.hipstr_gather_push(g, 0, 1, 400, b"x", 1) != 0 and b"order" in lib.hipstr_last_error()
    assert _pop(lib, g)[0] == 2                                  # stream 1 could still produce something smaller
    lib.hipstr_gather_end(g, 1)
    n = C.c_int64(); rc = lib.hipstr_gather_pop(g, None, None, None, C.create_string_buffer(1), 1, C.byref(n))
    assert rc == 1 and n.value == 3                              # too small: the record stays
    assert _pop(lib, g) == (0, (0, 1, 500, b"abc"))
    assert _pop(lib, g)[0] == 2
    lib.hipstr_gather_end(g, 0)
    assert _pop(lib, g)[0] == 3
    assert lib.hipstr_gather_push(g, 0, 2, 1, b"", 0) != 0       # ended
    lib.hipstr_gather_close(g)
