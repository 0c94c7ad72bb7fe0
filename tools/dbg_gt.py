"""Debug: genotype-call fields of the configs[3]-shape case, device vs oracle, printing where they differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hipstr_amd import capi
import test_config4_gpu as T

hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
o = capi.load_oracle()
A, S, V = 32, 1000, 8
pb = T._post_case(42, 3, A, S, (3, 8), haploid=[0, 1, 0])
h2a = np.tile((np.arange(A) // 2) % V, 3)
w = capi.run_gt_extract(o, "oracle_", pb, [V] * 3, h2a)
g = capi.run_gt_extract(hmm, "hipstr_", pb, [V] * 3, h2a)
for k in ("log_phased_post", "log_unphased_post", "hap_log_phased_post", "hap_log_unphased_post", "gl_diff"):
    d = np.abs(g[k] - w[k]); bad = np.where(~(d <= 1e-9 * np.maximum(1, np.abs(w[k]))))[0]
    print(k, "bad", len(bad), bad[:10])
    for s in bad[:5]:
        gs = np.sort(w["gls"][s])[::-1]; gg = np.sort(g["gls"][s])[::-1]
        print("  sample", s, "want", w[k][s], "got", g[k][s], "best_gt", w["best_gt"][s], g["best_gt"][s], "top gls want", gs[:3], "got", gg[:3])
        print("   max|dgl|", np.max(np.abs(w["gls"][s] - g["gls"][s])))
