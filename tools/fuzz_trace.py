"""Randomised parity sweep of the traceback (GPU vs oracle) over generator shapes.  usage: python tools/fuzz_trace.py [n_configs] [seed] [edges]
"edges": 63 ... 300 reads and 1 ... 160 alleles per locus (hundreds of requests per call: every column class in one launch), reads of
8 ... 40 bases."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hipstr_amd import capi
import util

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
bad = 0; total = 0
for c in range(n_cfg):
    os.environ["HIPSTR_SYNTH_IMPERFECT"] = str(float(rng.choice([0.0, 0.05, 0.5, 1.0])))
    os.environ["HIPSTR_SYNTH_INHERIT"] = str(int(rng.choice([0, 0, 1, 2, 3])))        # interruptions inherited from the reference allele
    kw = dict(n_loci=1, reads_per_locus=int(rng.integers(1, 30)), n_str_alleles=int(rng.integers(1, 25)), read_len=int(rng.integers(24, 251)),
              flank_len=int(rng.integers(8, 161)), str_bp=int(rng.integers(4, 121)), n_flank_opts=int(rng.integers(1, 4)), seed=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.15:          # long reads: sides of 385-1024 columns (the fill kernel's 8 / 12 / 16 columns per lane on dynamic LDS)
        kw.update(read_len=int(rng.integers(500, 1025)), flank_len=int(rng.integers(300, 620)), reads_per_locus=int(rng.integers(1, 8)), n_str_alleles=int(rng.integers(1, 5)))
    edges = len(sys.argv) > 3 and sys.argv[3] == "edges"
    if edges:
        kw.update(reads_per_locus=int(rng.choice([1, 63, 64, 65, 127, 129, 255, 257, 300])),          # (one locus per call: the oracle's trace takes one)
                  n_str_alleles=int(rng.choice([1, 2, 3, 32, 64, 65, 128, 160])), n_flank_opts=int(rng.choice([1, 1, 2])))
        if kw["n_str_alleles"] * kw["n_flank_opts"] ** 2 > 300: kw["n_flank_opts"] = 1
        if kw["read_len"] > 250: kw["reads_per_locus"] = min(kw["reads_per_locus"], 8)
        if rng.random() < 0.3: kw.update(read_len=int(rng.integers(8, 40)), flank_len=int(rng.integers(2, 20)), str_bp=int(rng.integers(4, 30)))
    if len(sys.argv) > 3 and sys.argv[3] == "tiny":      # round 6: two to a dozen copies of a period-1..3 motif (tools/fuzz_align.py "tiny")
        os.environ["HIPSTR_SYNTH_PERIOD"] = str(int(rng.choice([1, 1, 1, 2, 2, 3])))
        kw.update(str_bp=int(rng.integers(2, 14)), n_str_alleles=int(rng.integers(4, 30)), n_flank_opts=int(rng.choice([1, 2, 3])),
                  flank_len=int(rng.choice([3, 8, 20, 46, 65, 120])), read_len=int(rng.integers(20, 160)), reads_per_locus=int(rng.integers(4, 40)))
    else:
        os.environ["HIPSTR_SYNTH_PERIOD"] = "0"
    sb = capi.SynthBatch(**kw)
    _, seeds = capi.run_align(ora, "oracle_", sb.ptr)
    b = sb.ptr.contents
    hap_off = np.ctypeslib.as_array(b.hap_off, shape=(kw["n_loci"] + 1,)); read_off = np.ctypeslib.as_array(b.read_off, shape=(kw["n_loci"] + 1,))
    rr, aa = [], []
    for l in range(kw["n_loci"]):
        A = int(hap_off[l + 1] - hap_off[l])
        for r in range(int(read_off[l]), int(read_off[l + 1])):
            if seeds[r] >= 0:
                for k in rng.choice(A, size=min(A, 2), replace=False):
                    rr.append(r); aa.append(int(k))
    if not rr:
        continue
    h2r = capi.hap_aln_info(ora, "oracle_", sb.ptr)
    want = capi.run_trace(ora, "oracle_", sb.ptr, rr, aa, h2r, cap=1 << 21)
    got = capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 21)
    total += len(rr)
    if got != want:
        bad += 1
        q = next(i for i, (g, w) in enumerate(zip(got, want)) if g != w)
        print("MISMATCH", kw, os.environ["HIPSTR_SYNTH_IMPERFECT"], os.environ["HIPSTR_SYNTH_INHERIT"], "request", q, {f: (got[q][f], want[q][f]) for f in got[q] if got[q][f] != want[q][f]})
    if capi.hap_aln_info(hmm, "hipstr_", sb.ptr) != h2r:
        bad += 1; print("MISMATCH hap_aln_info", kw)
print("configs", n_cfg, "tracebacks", total, "mismatching configs", bad)
