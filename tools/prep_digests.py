#!/usr/bin/env python
"""Digests of the host preparation (hipstr_debug_prepare) over a corpus of generator shapes: a rewrite of prep.cpp must reproduce every
pool, offset and work item byte for byte.  `prep_digests.py save FILE` / `prep_digests.py check FILE` (HIPSTR_HMM_LIB selects the library)."""
import ctypes as C, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    ("p30", dict(n_loci=600, reads_per_locus=40, n_str_alleles=8, read_len=150, flank_len=35, str_bp=40), {}),
    ("c2", dict(n_loci=150, reads_per_locus=40, n_str_alleles=32), {}),
    ("ns", dict(n_loci=40, reads_per_locus=500, n_str_alleles=32), {}),
    ("c5", dict(n_loci=12, reads_per_locus=200, n_str_alleles=128, read_len=250, flank_len=110, str_bp=100), {}),
    ("c1", dict(n_loci=1, reads_per_locus=50, n_str_alleles=4), {}),
    ("imp1", dict(n_loci=100, reads_per_locus=40, n_str_alleles=16), {"HIPSTR_SYNTH_IMPERFECT": "1.0"}),
    ("imp.3", dict(n_loci=100, reads_per_locus=30, n_str_alleles=12, seed=5), {"HIPSTR_SYNTH_IMPERFECT": "0.3"}),
    ("flank2", dict(n_loci=80, reads_per_locus=21, n_str_alleles=6, n_flank_opts=2, seed=4, mask_rate=0.25), {}),
    ("flank3", dict(n_loci=40, reads_per_locus=33, n_str_alleles=5, n_flank_opts=3, seed=9, mask_rate=0.1), {"HIPSTR_SYNTH_IMPERFECT": "0.5"}),
    ("short", dict(n_loci=90, reads_per_locus=30, n_str_alleles=5, read_len=60, flank_len=25, str_bp=20, seed=13, mask_rate=0.2), {}),
    ("tiny", dict(n_loci=60, reads_per_locus=17, n_str_alleles=7, read_len=40, flank_len=12, str_bp=8, seed=21), {}),
    ("long", dict(n_loci=6, reads_per_locus=50, n_str_alleles=20, read_len=300, flank_len=140, str_bp=300, seed=31), {}),
    ("masked", dict(n_loci=70, reads_per_locus=25, n_str_alleles=10, seed=17, mask_rate=0.6), {}),
    # interruptions inherited from the reference allele: lists with three to six breaks (K-level descriptor slots, layout.h HS_SHAPE_PWK) and beyond
    ("inh2", dict(n_loci=60, reads_per_locus=30, n_str_alleles=16, seed=41), {"HIPSTR_SYNTH_INHERIT": "2"}),
    ("inh3", dict(n_loci=40, reads_per_locus=24, n_str_alleles=12, seed=43), {"HIPSTR_SYNTH_INHERIT": "3", "HIPSTR_SYNTH_IMPERFECT": "0.3"}),
]


def one(name):
    from hipstr_amd import capi
    kw = dict([c for c in CASES if c[0] == name][0][1])
    hmm = capi.load_hmm()
    sb = capi.SynthBatch(**kw)
    out = {}
    for threads in (1, 3):
        sec = C.c_double(); dig = C.c_uint64()
        assert hmm.hipstr_debug_prepare(sb.ptr, threads, C.byref(sec), C.byref(dig)) == 0, hmm.hipstr_last_error()
        out[str(threads)] = "%016x" % dig.value
    print(json.dumps(out))


def all_cases():
    res = {}
    for name, kw, env in CASES:
        e = dict(os.environ); e.update(env)
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "one", name], env=e, stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
        res[name] = json.loads(o.strip().splitlines()[-1])
    return res


if __name__ == "__main__":
    if sys.argv[1] == "one":
        one(sys.argv[2])
    elif sys.argv[1] == "save":
        json.dump(all_cases(), open(sys.argv[2], "w"), indent=1); print("saved")
    else:
        want = json.load(open(sys.argv[2])); got = all_cases()
        bad = [k for k in want if want[k] != got.get(k)]
        print("MISMATCH: " + ", ".join(bad) if bad else "all %d digests match" % len(want))
        sys.exit(1 if bad else 0)
