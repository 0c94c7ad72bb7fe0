"""Traceback throughput: hipstr_hmm_trace with one request per read (its source allele) on NS-shaped loci — per-locus calls and
one call for all loci — with the compiled reference's trace_optimal_aln (1 thread) beside it when oracle/_ref is present.
Prints one JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hipstr_amd import capi, shard
import util

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
sb = capi.SynthBatch(n_loci=n_loci, reads_per_locus=500, n_str_alleles=32, seed=1000)
seeds = np.zeros(sb.n_reads, np.int32)
hmm.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
src = sb.src_allele()
rr = [r for r in range(sb.n_reads) if seeds[r] >= 0]; aa = [int(src[r]) for r in rr]
whole = util.synth_to_batch(sb); a = whole.arrays
h2r = []
ones = []
for l in range(n_loci):
    one = shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1)); ones.append(one)
    h2r += util.synthetic_hap_to_ref(ora, one.ptr)
cap = 1 << 26
capi.run_trace(hmm, "hipstr_hmm_", whole.ptr, rr[:500], aa[:500], h2r, cap=cap, unpack=False)     # warm-up
tb = {}
capi.run_trace(hmm, "hipstr_hmm_", whole.ptr, rr, aa, h2r, cap=cap, timing=tb, unpack=False)
out = dict(metric="tracebacks_per_sec", value=len(rr) / tb["call_s"], loci=n_loci, requests=len(rr), ms_per_locus=1e3 * tb["call_s"] / n_loci,
           note="one C call for all loci, end to end: host prep + H2D + kernels + D2H + host replay/stitch")
# the same requests, one call per locus (what a per-locus caller sees)
tl = {}; n1 = 0
for l in range(min(n_loci, 8)):
    r0, r1 = int(a["read_off"][l]), int(a["read_off"][l + 1])
    sel = [i for i, r in enumerate(rr) if r0 <= r < r1]
    capi.run_trace(hmm, "hipstr_hmm_", ones[l].ptr, [rr[i] - r0 for i in sel], [aa[i] for i in sel], h2r[int(a["hap_off"][l]):int(a["hap_off"][l + 1])],
                   cap=1 << 22, timing=tl, unpack=False)
    n1 += len(sel)
out["per_locus_calls"] = n1 / tl["call_s"]
if os.path.exists(capi.REF_LIB):
    ref = capi.load_ref(); tr_ = {}
    r0, r1 = int(a["read_off"][0]), int(a["read_off"][1])
    sel = [i for i, r in enumerate(rr) if r0 <= r < r1][:200]
    capi.run_trace(ref, "ref_", ones[0].ptr, [rr[i] - r0 for i in sel], [aa[i] for i in sel], cap=1 << 22, timing=tr_, unpack=False)
    out["cpu_reference_1thread"] = len(sel) / tr_["call_s"]
print(json.dumps(out))
