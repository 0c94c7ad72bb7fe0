"""One-locus call latency as bench.py measures it (hipstr_hmm_process_reads on prepared arrays, median of 200) + the host-side buckets of the call."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipstr_amd import capi
hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
hmm.hipstr_debug_api_profile.restype = C.c_int
hmm.hipstr_debug_api_profile.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("HIPSTR_"))
for pp, aa in ((50, 4), (40, 32), (500, 32)):
    one = capi.SynthBatch(n_loci=1, reads_per_locus=pp, n_str_alleles=aa, seed=77)
    pr = np.zeros(one.n_out); sd = np.zeros(one.n_reads, np.int32)
    ts = []
    for _ in range(210):
        t1 = time.perf_counter()
        assert hmm.hipstr_hmm_process_reads(one.ptr, pr.ctypes.data_as(capi._f64p), sd.ctypes.data_as(capi._i32p)) == 0
        ts.append(time.perf_counter() - t1)
    print("%dx%d: median %.4f ms min %.4f  [%s]" % (pp, aa, 1e3*np.median(ts[10:]), 1e3*min(ts[10:]), tag))
    if pp == 40 and os.environ.get("LAT_BUCKETS"):
        names = (C.c_char_p*32)(); secs = (C.c_double*32)(); calls = (C.c_int64*32)()
        hmm.hipstr_debug_api_profile(1, 32, names, secs, calls)
        for _ in range(200):
            hmm.hipstr_hmm_process_reads(one.ptr, pr.ctypes.data_as(capi._f64p), sd.ctypes.data_as(capi._i32p))
        n = hmm.hipstr_debug_api_profile(0, 32, names, secs, calls)
        for i in range(n):
            if calls[i]: print("   %-40s %8.2f us/call" % (names[i].decode(), 1e6*secs[i]/200))
