"""GPU check of hipstr_hmm_trace against oracle_trace on seeded synthetic loci (run on the MI355X box)."""
import sys, numpy as np
sys.path.insert(0, '.')
from hipstr_amd import capi
sys.path.insert(0, 'tests')
import util
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
FIELDS = ("max_index", "hap_aln", "stutter_size", "str_seq", "flank_left", "flank_right", "flank_ins", "flank_del", "indels", "snps",
          "aln_start", "aln_stop", "cigar", "aln_str")
def run(alleles_per_read=3, **kw):
    sb = capi.SynthBatch(n_loci=1, **kw)
    _, seeds = capi.run_align(ora, "oracle_", sb.ptr)
    A = sb.n_out // sb.n_reads
    rng = np.random.default_rng(kw.get("seed", 0))
    rr, aa = [], []
    for r in range(sb.n_reads):
        if seeds[r] < 0: continue
        for k in rng.choice(A, size=min(A, alleles_per_read), replace=False):
            rr.append(r); aa.append(int(k))
    h2r = util.synthetic_hap_to_ref(ora, sb.ptr)
    want = capi.run_trace(ora, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 20)
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 20)
    bad = 0; maxd = 0.0
    for q, (g, w) in enumerate(zip(got, want)):
        maxd = max(maxd, abs(g["ll"] - w["ll"]))
        for f in FIELDS:
            if g[f] != w[f]:
                bad += 1
                if bad <= 5: print("  req", q, "read", rr[q], "allele", aa[q], f, "got", g[f], "want", w[f])
    print(kw, "requests", len(rr), "field mismatches", bad, "max |dLL|", maxd, flush=True)
run(reads_per_locus=50, n_str_alleles=4, seed=1)
run(reads_per_locus=40, n_str_alleles=8, seed=7)
run(reads_per_locus=30, n_str_alleles=5, n_flank_opts=2, seed=11)
run(reads_per_locus=20, n_str_alleles=16, read_len=250, flank_len=110, str_bp=100, seed=5)
run(reads_per_locus=60, n_str_alleles=12, read_len=100, flank_len=35, str_bp=30, seed=3)
run(reads_per_locus=30, n_str_alleles=6, read_len=250, flank_len=160, str_bp=60, seed=9)
