import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from hipstr_amd import capi
from nw_cases import nw_pairs
hmm=capi.load_hmm(); ora=capi.load_oracle()
assert hmm.hipstr_hmm_init(0)==0
bad=tot=0
for seed in range(6):
    for pen in (False, True):
        pairs=nw_pairs(seed, n=60, read_len=(1,256) if seed==5 else (30,150))
        a=capi.run_nw(ora,"oracle_",pairs,pen); b=capi.run_nw(hmm,"hipstr_",pairs,pen)
        for i,(x,y) in enumerate(zip(a,b)):
            tot+=1
            if x!=y:
                bad+=1
                if bad<=3: print("MISMATCH seed",seed,pen,i,"\n",x,"\n",y)
print("pairs",tot,"mismatches",bad)
# throughput: 500 unique reads of 150 bp against 300 bp windows (one NS locus worth), x64 loci
pairs=nw_pairs(99, n=8000, ref_len=(290,310), read_len=(140,150))
t={}; capi.run_nw(hmm,"hipstr_",pairs[:64],False,unpack=False)
capi.run_nw(hmm,"hipstr_",pairs,False,unpack=False,timing=t)
cells=sum(len(r)*len(q) for r,q in pairs)
print("GPU: %d pairs %.1f ms -> %.0f pairs/s, %.2e cells/s" % (len(pairs), 1e3*t["call_s"], len(pairs)/t["call_s"], cells/t["call_s"]))
t2={}; capi.run_nw(ora,"oracle_",pairs[:400],False,unpack=False,timing=t2)
print("oracle 1 thread: %.0f pairs/s" % (400/t2["call_s"]))
import os
if os.path.exists(capi.REF_LIB):
    t3={}; capi.run_nw(capi.load_ref(),"ref_",pairs[:400],False,unpack=False,timing=t3)
    print("reference 1 thread: %.0f pairs/s" % (400/t3["call_s"]))
