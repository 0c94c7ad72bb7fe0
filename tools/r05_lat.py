"""One-locus call latency (hipstr_hmm_process_reads, median of 40) for the shapes the driver line quotes, and the kernel phases of a resident pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipstr_amd import capi
hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
ora = capi.load_oracle()
for P, A in ((50, 4), (40, 32), (500, 32)):
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=P, n_str_alleles=A, seed=3)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    want, ws = capi.run_align(ora, "oracle_", sb.ptr)
    ok = np.array_equal(got, want) and np.array_equal(gs, ws)
    ts = []
    for _ in range(40):
        t = time.perf_counter(); capi.run_align(hmm, "hipstr_hmm_", sb.ptr); ts.append(time.perf_counter() - t)
    print("one-shot process_reads %dx%d: median %.3f ms min %.3f  (== oracle: %s)  FLANK_SYSTOLIC=%s" % (P, A, 1e3*np.median(ts), 1e3*min(ts), ok, os.environ.get("HIPSTR_FLANK_SYSTOLIC", "default")))
