import sys, numpy as np
sys.path.insert(0, '.')
from hipstr_amd import capi
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
def run(**kw):
    sb = capi.SynthBatch(**kw)
    got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
    want, ws = capi.run_align(ora, "oracle_", sb.ptr)
    both = np.isfinite(got) & np.isfinite(want)
    nanmis = int((np.isfinite(got) != np.isfinite(want)).sum())
    d = np.abs(np.where(both, got - want, 0))
    bad = np.nonzero(d > 0)[0]
    print(kw, "n", sb.n_out, "seeds_eq", np.array_equal(gs, ws), "nanmismatch", nanmis, "nbad", len(bad), "max", d.max() if len(d) else 0, flush=True)
    for i in bad[:6]:
        l = int(np.searchsorted(sb.out_off, i, side='right') - 1)
        print("   locus", l, "idx", int(i - sb.out_off[l]), "got %.12f want %.12f diff %.3e" % (got[i], want[i], got[i]-want[i]))
run(n_loci=2, reads_per_locus=50, n_str_alleles=4, seed=1)
run(n_loci=20, reads_per_locus=20, n_str_alleles=8, seed=7)
run(n_loci=10, reads_per_locus=10, n_str_alleles=5, n_flank_opts=2, seed=11)
run(n_loci=10, reads_per_locus=12, n_str_alleles=6, n_flank_opts=3, seed=13, mask_rate=0.3)
run(n_loci=3, reads_per_locus=10, n_str_alleles=16, read_len=250, flank_len=110, str_bp=100, seed=5)
run(n_loci=8, reads_per_locus=30, n_str_alleles=32, seed=21)
run(n_loci=6, reads_per_locus=30, n_str_alleles=12, read_len=100, flank_len=35, str_bp=30, seed=3)
