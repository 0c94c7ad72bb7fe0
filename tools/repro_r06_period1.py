"""The two mismatching configurations of tools/r06_fuzz_fresh.sh 660000 (period 1, every alt allele interrupted).  usage: python tools/repro_r06_period1.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipstr_amd import capi
hmm = capi.load_hmm(); ora = capi.load_oracle()
assert hmm.hipstr_hmm_init(0) == 0
CASES = [dict(n_loci=2, reads_per_locus=13, n_str_alleles=35, read_len=216, flank_len=46, str_bp=17, n_flank_opts=2, seed=650757509, mask_rate=0.0),
         dict(n_loci=2, reads_per_locus=65, n_str_alleles=63, read_len=156, flank_len=65, str_bp=18, n_flank_opts=1, seed=717277815, mask_rate=0.3)]
for kw in CASES:
    os.environ["HIPSTR_SYNTH_IMPERFECT"] = "1.0"; os.environ["HIPSTR_SYNTH_INHERIT"] = "0"; os.environ["HIPSTR_SYNTH_PERIOD"] = "1"
    sb = capi.SynthBatch(**kw)
    want, ws = capi.run_align(ora, "oracle_", sb.ptr, fill=-3.25)
    b = sb.ptr.contents
    nl = b.n_loci
    hap_off = np.ctypeslib.as_array(b.hap_off, shape=(nl + 1,)); read_off = np.ctypeslib.as_array(b.read_off, shape=(nl + 1,))
    for env in ({}, {"HIPSTR_STR_GROUP": "0"}):
        for k in ("HIPSTR_STR_GROUP",): os.environ.pop(k, None)
        os.environ.update(env)
        got, gs = capi.run_align(hmm, "hipstr_hmm_", sb.ptr, fill=-3.25)
        d = np.abs(got - want)
        print(kw["seed"], env, "seeds equal", np.array_equal(gs, ws), "nbad", int((d > 0).sum()), "max", d.max())
        if (d > 0).any():
            # which (locus, read, allele)
            out_off = 0
            for l in range(nl):
                A = int(hap_off[l + 1] - hap_off[l]); R = int(read_off[l + 1] - read_off[l])
                blk = d[out_off:out_off + R * A].reshape(R, A)
                if (blk > 0).any():
                    rr, aa = np.nonzero(blk > 0)
                    print("  locus", l, "A", A, "R", R, "bad alleles", sorted(set(aa.tolist())), "bad reads", sorted(set(rr.tolist()))[:20])
                    nopts = np.ctypeslib.as_array(b.blk_nopts, shape=(3 * nl,))[3 * l:3 * l + 3]
                    # the STR options of the locus
                    ob = int(np.ctypeslib.as_array(b.blk_nopts, shape=(3 * nl,))[:3 * l].sum())
                    opt_off = np.ctypeslib.as_array(b.opt_off, shape=(int(np.ctypeslib.as_array(b.blk_nopts, shape=(3 * nl,)).sum()) + 1,))
                    seq = C.string_at(b.seq, int(opt_off[-1])) if False else None
                    import ctypes as C
                    raw = C.string_at(b.seq, int(opt_off[-1]))
                    so = ob + int(nopts[0])
                    strs = [raw[opt_off[so + i]:opt_off[so + i + 1]].decode() for i in range(int(nopts[1]))]
                    print("  nopts", nopts.tolist(), "STR options of the bad alleles:", [(a, strs[(a // int(nopts[2])) % int(nopts[1])]) for a in sorted(set(aa.tolist()))][:12])
                    r0 = rr[0]; a0 = aa[0]
                    print("  e.g. read", r0, "allele", a0, "got", got[out_off + r0 * A + a0], "want", want[out_off + r0 * A + a0])
                out_off += R * A
    os.environ.pop("HIPSTR_STR_GROUP", None)
