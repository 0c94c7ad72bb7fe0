import os, sys, numpy as np
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
from hipstr_amd import capi
import util
hmm=capi.load_hmm(); ora=capi.load_oracle(); assert hmm.hipstr_hmm_init(0)==0
for period, imp, n_str in [(9,"1.0",125),(9,"0.0",125),(4,"1.0",300),(4,"0.0",480),(2,"0.3",1000)]:
    os.environ.update(HIPSTR_SYNTH_PERIOD=str(period), HIPSTR_SYNTH_IMPERFECT=imp, HIPSTR_SYNTH_INHERIT="0")
    sb=capi.SynthBatch(n_loci=1, reads_per_locus=40, n_str_alleles=n_str, read_len=67 if period==9 else 150, flank_len=95, str_bp=10, seed=564467067)
    nopt=np.ctypeslib.as_array(sb.ptr.contents.blk_nopts,shape=(3,)); lens=np.diff(np.ctypeslib.as_array(sb.ptr.contents.opt_off,shape=(int(nopt.sum())+1,)))[nopt[0]:nopt[0]+nopt[1]]
    _,ws=capi.run_align(ora,"oracle_",sb.ptr)
    order=np.argsort(lens,kind="stable")
    rr=[r for r in range(sb.n_reads) if ws[r]>=0][:10]; aa=[int(order[-1-(i%4)]) for i in range(len(rr))]
    h2r=util.synthetic_hap_to_ref(ora,sb.ptr)
    try:
        got=capi.run_trace(hmm,"hipstr_hmm_",sb.ptr,rr,aa,h2r,cap=1<<23)
        want=capi.run_trace(ora,"oracle_",sb.ptr,rr,aa,h2r,cap=1<<23)
        util.assert_traces_equal(got,want,"long")
        print("period",period,imp,"maxB",int(lens.max()),"traces",len(rr),"OK")
    except Exception as e:
        print("period",period,imp,"maxB",int(lens.max()),"FAIL",str(e)[:300])
