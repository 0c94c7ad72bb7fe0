"""Where a traceback call's time goes (HIPSTR_TRACE_TIMING=1 prints the library's own split): the bench's pipeline.traceback case —
every seeded read of 32 north-star loci (500 reads x 30 alleles) against its source allele.  usage: python tools/r06_trace_timing.py [threads]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["HIPSTR_TRACE_TIMING"] = "1"
if len(sys.argv) > 1: os.environ["HIPSTR_HOST_THREADS"] = sys.argv[1]
import numpy as np
from hipstr_amd import capi
hmm = capi.load_hmm()
P, nl = 500, 32
sb = capi.SynthBatch(n_loci=nl, reads_per_locus=P, n_str_alleles=30, seed=20260928)
seeds = np.zeros(sb.n_reads, np.int32)
hmm.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
src = sb.src_allele()
rr = [r for r in range(nl * P) if seeds[r] >= 0][:20000]; aa = [int(src[r]) for r in rr]
h2r = capi.hap_aln_info(hmm, "hipstr_", sb.ptr, cap=1 << 26)
capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr[:256], aa[:256], h2r, cap=1 << 24, unpack=False)
for rep in range(3):
    t = {}
    capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 24, timing=t, unpack=False)
    print("requests %d call %.2f ms -> %.0f k/s" % (len(rr), 1e3 * t["call_s"], len(rr) / t["call_s"] / 1e3), flush=True)
