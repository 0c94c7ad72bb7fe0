import sys, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from hipstr_amd import capi
from em_cases import em_case
hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
kw = em_case(5, n_loci=3000, samples=(100, 100), reads_per_sample=(6, 6), allele_counts=[32]*3000)
import time
for i in range(2):
    t=time.time(); got = capi.run_em(hmm, "hipstr_", **kw); print("em", time.time()-t, got[2].mean())
x=np.random.default_rng(1).uniform(-40,0,16_000_000); y=np.empty_like(x)
import ctypes as C
hmm.hipstr_debug_cr_math(0, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), x.size)
x=np.random.default_rng(1).uniform(1,1000,16_000_000)
hmm.hipstr_debug_cr_math(1, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), x.size)
