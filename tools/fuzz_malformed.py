#!/usr/bin/env python
"""Malformed batches at the C-ABI (host only, no device): the boundary takes the caller's arrays as they are, so the host preparation
(prepare_batch / check_locus, prep.cpp) must REFUSE a batch whose offsets, counts, lengths or characters make no sense — with an error
message, not with a crash or a hang.  A valid generator batch gets one random corruption at a time (an offset negative / huge / decreasing,
a count zero / negative / huge, a CIGAR length or character off, a period out of range, ...) and goes through hipstr_debug_prepare in a
child process; the parent reports crashes (signals) and timeouts with the corruption that caused them.   usage: tools/fuzz_malformed.py [n] [seed]
Child mode (internal): tools/fuzz_malformed.py --child <seed> <first> <count>"""
import os, subprocess, sys
import ctypes as C
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))

INT_FIELDS = ["blk_start", "blk_end", "blk_nopts", "period", "opt_off", "hap_off", "read_off", "base_off", "read_start", "cigar_off", "cigar_len"]
BYTE_FIELDS = ["seq", "bases", "quals", "cigar_op"]


def corrupt(a, rng):
    """One corruption of the arrays dict `a` (in place); returns its description."""
    kind = int(rng.integers(6))
    if kind <= 2:
        f = str(rng.choice(INT_FIELDS)); v = a[f]
        i = int(rng.integers(v.size)); old = int(v[i])
        new = int(rng.choice([-1, 0, 1, old - 1, old + 1, old * 2 + 3, 2 ** 30, -2 ** 31, 10 ** 6]))
        v[i] = new
        return "%s[%d]: %d -> %d" % (f, i, old, new)
    if kind == 3:
        f = str(rng.choice(BYTE_FIELDS)); v = bytearray(a[f])
        if len(v) <= 1: return "none"
        i = int(rng.integers(len(v) - 1)); old = v[i]; v[i] = int(rng.choice([0, 1, ord('N'), ord('n'), ord('a'), ord('='), ord('X'), ord('S'), ord('H'), ord('?'), 255, ord('!') - 1, ord('~') + 1]))
        a[f] = bytes(v)
        return "%s[%d]: %d -> %d" % (f, i, old, v[i])
    if kind == 4:
        i = int(rng.integers(a["stutter"].size)); old = float(a["stutter"][i]); new = float(rng.choice([0.0, 1.0, -0.5, 2.0, np.nan, np.inf, 1e-320]))
        a["stutter"][i] = new
        return "stutter[%d]: %g -> %g" % (i, old, new)
    f = str(rng.choice(["opt_off", "hap_off", "read_off", "base_off", "cigar_off"])); v = a[f]
    if v.size < 3: return "none"
    i = int(rng.integers(1, v.size - 1)); v[i], v[i + 1] = v[i + 1], v[i]
    return "%s[%d] <-> [%d]" % (f, i, i + 1)


def child(seed, first, count):
    from hipstr_amd import capi, shard
    import fuzz_mixed
    hmm = capi.load_hmm()
    hmm.hipstr_debug_prepare.restype = C.c_int
    rng0 = np.random.default_rng(seed)
    base = []
    for kw in (dict(n_loci=3, reads_per_locus=6, n_str_alleles=4, seed=11), dict(n_loci=2, reads_per_locus=9, n_str_alleles=7, n_flank_opts=2, read_len=60, flank_len=25, seed=12, mask_rate=0.3)):
        sb = capi.SynthBatch(**kw); base.append(fuzz_mixed.arrays_from_ptr(sb.ptr)); sb.close()
    for k in range(first, first + count):
        rng = np.random.default_rng([seed, k])
        a = {f: (v.copy() if isinstance(v, np.ndarray) else v) for f, v in base[k % 2].items()}
        what = corrupt(a, rng)
        if rng.random() < 0.3: what += " ; " + corrupt(a, rng)
        print("case %d: %s" % (k, what), flush=True)
        b = shard.batch_from_arrays(a)
        sec = C.c_double(); dig = C.c_uint64()
        rc = hmm.hipstr_debug_prepare(b.ptr, 1, C.byref(sec), C.byref(dig))
        print("  rc %d %s" % (rc, hmm.hipstr_last_error().decode()[:90] if rc else ""), flush=True)
    print("child done", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    step = 50; crashes = []; refused = accepted = 0
    k = 0
    while k < n:
        cnt = min(step, n - k)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(seed), str(k), str(cnt)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               universal_newlines=True, timeout=120)
            out, rc = r.stdout, r.returncode
        except subprocess.TimeoutExpired as e:
            out, rc = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""), -999
        lines = [l for l in out.splitlines() if l.startswith("case ") or l.startswith("  rc")]
        refused += sum(1 for l in lines if l.startswith("  rc 1")); accepted += sum(1 for l in lines if l.startswith("  rc 0"))
        if "child done" in out:
            k += cnt; continue
        last = [l for l in lines if l.startswith("case ")]
        done = len([l for l in lines if l.startswith("  rc")])
        crashes.append((rc, last[-1] if last else out[-300:]))
        print("CRASH (child rc %d) at %s" % (rc, last[-1] if last else out[-300:]), flush=True)
        k += done + 1                      # past the case that killed the child
    print("cases %d refused %d accepted %d crashes/timeouts %d" % (n, refused, accepted, len(crashes)))
    return len(crashes)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        sys.exit(1 if main() else 0)
