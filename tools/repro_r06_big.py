"""Round 6: the mismatch tools/fuzz_align.py "big" found (loci with interrupted alleles of more than 512 bp) — a matrix of variants around it."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipstr_amd import capi
hmm=capi.load_hmm(); ora=capi.load_oracle(); assert hmm.hipstr_hmm_init(0)==0
base={'n_loci': 1, 'reads_per_locus': 40, 'n_str_alleles': 125, 'read_len': 67, 'flank_len': 95, 'str_bp': 10, 'n_flank_opts': 1, 'seed': 564467067, 'mask_rate': 0.0}
for period, imp, inh, extra in [(9,"1.0","0",{}),(9,"0.0","0",{}),(4,"1.0","0",{"n_str_alleles":250}),(4,"0.0","0",{"n_str_alleles":250}),(9,"1.0","0",{"read_len":150}),(9,"1.0","0",{"n_str_alleles":70}),
                                (6,"1.0","0",{"n_str_alleles":180}),(7,"1.0","0",{"n_str_alleles":160}),(4,"0.3","2",{"n_str_alleles":250}), (9,"1.0","0",{"reads_per_locus":3})]:
    os.environ["HIPSTR_SYNTH_IMPERFECT"]=imp; os.environ["HIPSTR_SYNTH_INHERIT"]=inh; os.environ["HIPSTR_SYNTH_PERIOD"]=str(period)
    kw=dict(base); kw.update(extra)
    sb=capi.SynthBatch(**kw)
    want,ws=capi.run_align(ora,"oracle_",sb.ptr,fill=-3.25)
    for env in ({}, {"HIPSTR_STR_GROUP":"0"}, {"HIPSTR_STR_GROUP_PW":"0"}):
        for k in ("HIPSTR_STR_GROUP","HIPSTR_STR_GROUP_PW"): os.environ.pop(k, None)
        os.environ.update(env)
        got,gs=capi.run_align(hmm,"hipstr_hmm_",sb.ptr,fill=-3.25)
        A=sb.n_out//sb.n_reads
        bad=np.argwhere((got!=want).reshape(sb.n_reads,A))
        b=sb.ptr.contents
        nopt=np.ctypeslib.as_array(b.blk_nopts,shape=(3,)); opt_off=np.ctypeslib.as_array(b.opt_off,shape=(int(nopt.sum())+1,)); lens=np.diff(opt_off)[nopt[0]:nopt[0]+nopt[1]]
        print("period",period,"imperfect",imp,"inherit",inh,extra,env,"A",A,"maxB",int(lens.max()),"nbad",len(bad), "shortest bad allele bp", int(lens[bad[:,1]].min()) if len(bad) else None, "bad reads", len(set(bad[:,0].tolist())), flush=True)
