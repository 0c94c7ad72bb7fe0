import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, time
from hipstr_amd import capi
from em_cases import em_case
hmm=capi.load_hmm(); ora=capi.load_oracle()
assert hmm.hipstr_hmm_init(0)==0
for seed in range(5):
    kw=em_case(seed, n_loci=6)
    a=capi.run_em(ora,"oracle_",**kw); b=capi.run_em(hmm,"hipstr_",**kw)
    print(seed, a[2], b[2], np.array_equal(a[0],b[0]), np.abs(a[1]-b[1]).max(), np.abs(a[3]-b[3]).max())
kw=em_case(100, n_loci=200, samples=(80,100), reads_per_sample=(4,8))
t=time.time(); b=capi.run_em(hmm,"hipstr_",**kw); t1=time.time()-t
print("200 loci x ~100 samples:", t1, "s", "iters", b[2].sum(), "trained", b[0].sum())
kw2={k:(v[:5] if k in("period","n_samples","haploid") else v) for k,v in kw.items()}
kw2["read_off"]=kw["read_off"][:6]; n=kw2["read_off"][-1]
for k in ("sample_label","num_bps","log_p1","log_p2"): kw2[k]=kw[k][:n]
t=time.time(); a=capi.run_em(ora,"oracle_",**kw2); t2=time.time()-t
print("oracle 5 loci:", t2, "s ->", t2/5*200, "s per 200; params diff", np.abs(a[1]-b[1][:5]).max(), "iters", a[2], b[2][:5])
