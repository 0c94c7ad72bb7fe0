import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipstr_amd import capi
hmm = capi.load_hmm(); ora = capi.load_oracle(); hmm.hipstr_hmm_init(0)
# the refusals of the big mode: message in full
for env, kw in (((0.0,0,0), dict(n_loci=2, reads_per_locus=20, n_str_alleles=1000, read_len=146, flank_len=63, str_bp=42, n_flank_opts=1, seed=576076085, mask_rate=0.0)),
                ((0.0,0,0), dict(n_loci=1, reads_per_locus=8, n_str_alleles=500, read_len=49, flank_len=128, str_bp=87, n_flank_opts=1, seed=183655443, mask_rate=0.3))):
    for imp in (0.0, 0.05, 0.3, 1.0):
      for per in (0, 1, 7, 8, 9):
        os.environ["HIPSTR_SYNTH_IMPERFECT"] = str(imp); os.environ["HIPSTR_SYNTH_INHERIT"] = "0"; os.environ["HIPSTR_SYNTH_PERIOD"] = str(per)
        try:
            sb = capi.SynthBatch(**kw)
            capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
        except Exception as e:
            print("refused", kw["seed"], imp, per, str(e)[:300]); 
