import sys, time; sys.path.insert(0,'.')
import numpy as np
from hipstr_amd import capi
hmm=capi.load_hmm(); assert hmm.hipstr_hmm_init(0)==0
for P,A in ((50,4),(500,32)):
    sb=capi.SynthBatch(n_loci=1, reads_per_locus=P, n_str_alleles=A, seed=3)
    capi.run_align(hmm,"hipstr_hmm_",sb.ptr)
    ts=[]
    for _ in range(20):
        t=time.perf_counter(); capi.run_align(hmm,"hipstr_hmm_",sb.ptr); ts.append(time.perf_counter()-t)
    print("one-shot process_reads %dx%d: median %.3f ms min %.3f" % (P,A,1e3*np.median(ts),1e3*min(ts)))
    dev=hmm.hipstr_hmm_upload(sb.ptr)
    ts=[]
    for _ in range(20):
        t=time.perf_counter(); hmm.hipstr_hmm_align(dev,None); p=np.zeros(sb.n_out); s=np.zeros(sb.n_reads,np.int32)
        hmm.hipstr_hmm_fetch(dev,p.ctypes.data_as(capi._f64p),s.ctypes.data_as(capi._i32p)); ts.append(time.perf_counter()-t)
    print("   resident align+fetch: median %.3f ms" % (1e3*np.median(ts)))
    ts=[]
    for _ in range(10):
        t=time.perf_counter(); d2=hmm.hipstr_hmm_upload(sb.ptr); t1=time.perf_counter(); hmm.hipstr_hmm_free(d2); ts.append((t1-t, time.perf_counter()-t1))
    print("   upload %.3f ms, free %.3f ms" % (1e3*np.median([a for a,b in ts]), 1e3*np.median([b for a,b in ts])))
    hmm.hipstr_hmm_free(dev)
