import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from hipstr_amd import capi
import util
hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
sb = capi.SynthBatch(n_loci=1, reads_per_locus=100, n_str_alleles=8, seed=3)
seeds = np.zeros(sb.n_reads, np.int32); hmm.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
src = sb.src_allele(); rr = [r for r in range(sb.n_reads) if seeds[r] >= 0]; aa = [int(src[r]) for r in rr]
h2r = capi.hap_aln_info(hmm, "hipstr_", sb.ptr, cap=1 << 22)
capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 20, unpack=False)
