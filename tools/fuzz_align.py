"""Randomised parity sweep of the forward path (GPU vs oracle) over generator shapes: read length, flank length, STR size, allele
count, flank options, masks, share of interrupted repeats.  usage: python tools/fuzz_align.py [n_configs] [seed] [edges]
"edges": reads per locus and alleles per locus around the kernels' packing sizes (64 lanes, 256-lane workgroups and their multiples), very
short reads and flanks.  "tiny" (round 6): two to a dozen copies of a period-1..3 motif.  "big" (round 6): 160 ... 1000 candidate haplotypes per locus, 600 ... 5000 reads per locus.  Every mode: a third of
the configurations with the STR periods the generator's weights never draw (1, 7, 8, 9)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipstr_amd import capi

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
bad = 0; total = 0
for c in range(n_cfg):
    os.environ["HIPSTR_SYNTH_IMPERFECT"] = str(float(rng.choice([0.0, 0.05, 0.3, 1.0])))
    os.environ["HIPSTR_SYNTH_INHERIT"] = str(int(rng.choice([0, 0, 1, 2, 3])))        # interruptions inherited from the reference allele (round 4)
    os.environ["HIPSTR_SYNTH_PERIOD"] = str(int(rng.choice([0, 0, 0, 0, 1, 7, 8, 9])))      # round 6: the periods the generator's weights never draw (stutter_model.h:38: 1..9)
    read_len = int(rng.integers(24, 251))
    kw = dict(n_loci=int(rng.integers(1, 5)), reads_per_locus=int(rng.integers(1, 40)), n_str_alleles=int(rng.integers(1, 41)), read_len=read_len,
              flank_len=int(rng.integers(8, 161)), str_bp=int(rng.integers(4, 121)), n_flank_opts=int(rng.integers(1, 4)),
              seed=int(rng.integers(1, 1 << 30)), mask_rate=float(rng.choice([0.0, 0.0, 0.3])))
    if len(sys.argv) > 3 and sys.argv[3] == "edges":
        kw.update(n_loci=int(rng.integers(1, 3)), reads_per_locus=int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300])) + int(rng.integers(0, 2)),
                  n_str_alleles=int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 100, 128, 129, 160])), n_flank_opts=int(rng.choice([1, 1, 2, 3])))
        if kw["n_str_alleles"] * kw["n_flank_opts"] ** 2 > 400: kw["n_flank_opts"] = 1
        if kw["reads_per_locus"] * kw["n_str_alleles"] * kw["n_flank_opts"] ** 2 > 40000: kw["reads_per_locus"] = max(1, 40000 // (kw["n_str_alleles"] * kw["n_flank_opts"] ** 2))
        if rng.random() < 0.3: kw.update(read_len=int(rng.integers(8, 40)), flank_len=int(rng.integers(2, 20)), str_bp=int(rng.integers(4, 30)))
    if len(sys.argv) > 3 and sys.argv[3] == "big":
        # round 6: up to MAX_TOTAL_HAPLOTYPES = 1000 candidate haplotypes (genotyper_bam_processor.h:110) and thousands of reads per locus (configs[3]: 5000)
        nf = int(rng.choice([1, 2, 3, 4, 5])); ns = int(rng.choice([160, 200, 250, 333, 500, 1000])) // (nf * nf)
        kw.update(n_loci=int(rng.integers(1, 3)), n_flank_opts=nf, n_str_alleles=max(2, ns), reads_per_locus=int(rng.choice([3, 8, 20, 40])))
        if rng.random() < 0.35:
            kw.update(n_flank_opts=1, n_str_alleles=int(rng.choice([8, 16, 32])), reads_per_locus=int(rng.choice([601, 1000, 1023, 1025, 2500, 5000])), n_loci=1)
    if len(sys.argv) > 3 and sys.argv[3] == "tiny":
        # round 6 (after the two-copy homopolymer bug): the smallest alleles the generator makes — two to a dozen copies of a period-1..3 motif,
        # every flank option count, short and long flanks, many alleles so that the shortest ones are all there
        os.environ["HIPSTR_SYNTH_PERIOD"] = str(int(rng.choice([1, 1, 1, 2, 2, 3])))
        kw.update(str_bp=int(rng.integers(2, 14)), n_str_alleles=int(rng.integers(4, 30)), n_flank_opts=int(rng.choice([1, 2, 3])),
                  flank_len=int(rng.choice([3, 8, 20, 46, 65, 120])), read_len=int(rng.integers(20, 160)), reads_per_locus=int(rng.integers(4, 70)))
    try:
        sb = capi.SynthBatch(**kw)
        want, ws = capi.run_align(ora, "oracle_", sb.ptr, fill=-3.25)
    except Exception as e:           # shapes the generator or the oracle refuses are not interesting here
        print("skip", kw, str(e)[:80]); continue
    try:
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
    except RuntimeError as e:
        print("GPU refused", kw, str(e)[:100]); continue
    total += got.size
    ok = np.array_equal(gs, ws) and np.array_equal(got, want)
    if not ok:
        bad += 1
        d = np.abs(got - want)
        print("MISMATCH", kw, os.environ["HIPSTR_SYNTH_IMPERFECT"], os.environ["HIPSTR_SYNTH_INHERIT"], "period", os.environ["HIPSTR_SYNTH_PERIOD"], "n", got.size, "nbad", int((d > 0).sum()), "max", d.max())
print("configs", n_cfg, "alignments", total, "mismatching configs", bad)
