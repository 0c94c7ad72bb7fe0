"""Traceback throughput: hipstr_hmm_trace (one request per read, its source allele) on NS-shaped loci, with the compiled
reference's trace_optimal_aln (1 thread) beside it when oracle/_ref is present.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from hipstr_amd import capi
import util

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
tot = 0; t_gpu = 0.0; t_ref = 0.0; n_ref = 0; tg = {}; tr_ = {}
ref = capi.load_ref() if os.path.exists(capi.REF_LIB) else None
for l in range(n_loci):
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=500, n_str_alleles=32, seed=1000 + l)
    seeds = np.zeros(sb.n_reads, np.int32)
    hmm.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
    src = sb.src_allele()
    rr = [r for r in range(sb.n_reads) if seeds[r] >= 0]; aa = [int(src[r]) for r in rr]
    h2r = util.synthetic_hap_to_ref(ora, sb.ptr)
    if l == 0:
        capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 22)      # warm-up
    t0 = time.perf_counter()
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 22, timing=tg)
    t_gpu += time.perf_counter() - t0; tot += len(rr)
    if ref is not None and l < 2:
        t0 = time.perf_counter()
        capi.run_trace(ref, "ref_", sb.ptr, rr[:100], aa[:100], cap=1 << 22, timing=tr_)
        t_ref += time.perf_counter() - t0; n_ref += 100
out = dict(metric="tracebacks_per_sec", value=tot / tg["call_s"], with_python_unpacking=tot / t_gpu, loci=n_loci, requests=tot, ms_per_locus=1e3 * tg["call_s"] / n_loci,
           note="value = the C call end to end: host prep + H2D + kernels + D2H + host replay/stitch")
if n_ref:
    out["cpu_reference_1thread"] = n_ref / tr_["call_s"]
print(json.dumps(out))
