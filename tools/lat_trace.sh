#!/bin/bash
# lat_trace.sh [align|trace] — kernel timeline of ONE-locus calls (forward: 40 reads x 32 alleles; trace: 100 requests): what the 0.3 ms of a hipstr_hmm_process_reads call are made of.
# Prints, per kernel, its average duration and the average gap to the previous kernel's end within a call (from rocprofv3's kernel trace).
R=$(pwd); O=$R/gpurun_out/lat; rm -rf $O; mkdir -p $O
cat > $O/one.py <<'PY'
import os, sys, time; sys.path.insert(0, sys.argv[1])
import numpy as np
from hipstr_amd import capi
hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
mode = sys.argv[2] if len(sys.argv) > 2 else "align"
if mode == "align":
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=40, n_str_alleles=32, seed=3)
    call = lambda: capi.run_align(hmm, "hipstr_hmm_", sb.ptr)
else:      # one traceback call of a locus: 100 reads, each against its source allele
    sys.path.insert(0, sys.argv[1] + "/tests")
    import util
    sb = capi.SynthBatch(n_loci=1, reads_per_locus=100, n_str_alleles=8, seed=3)
    seeds = np.zeros(sb.n_reads, np.int32); hmm.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
    src = sb.src_allele(); rr = [r for r in range(sb.n_reads) if seeds[r] >= 0]; aa = [int(src[r]) for r in rr]
    h2r = capi.hap_aln_info(hmm, "hipstr_", sb.ptr, cap=1 << 22)
    call = lambda: capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 20, unpack=False)
if os.environ.get("LAT_BUSY"):      # keep the device busy (and its clocks up) from a second thread while the calls are timed
    import threading, torch
    stop = [False]
    def spin():
        a = torch.randn(4096, 4096, device="cuda"); s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            while not stop[0]:
                for _ in range(20): a = (a @ a).clamp_(-1, 1)
                s2.synchronize()
    th = threading.Thread(target=spin, daemon=True); th.start(); time.sleep(1.0)
for _ in range(5): call()
ts = []
for _ in range(40):
    t = time.perf_counter(); call(); ts.append(time.perf_counter() - t); time.sleep(0.002)
print("%s, one locus per call: median %.3f ms min %.3f" % (mode, 1e3*np.median(ts), 1e3*min(ts)))
PY
cd /tmp && export TMPDIR=/tmp
python $O/one.py $R ${1:-align}
rocprofv3 --kernel-trace -d $O/trace -o v -- python $O/one.py $R ${1:-align} > $O/trace.log 2>&1
python - $(find $O/trace -name '*results.db' | head -1) <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("kernels") or t == "kernels"]
rows = list(db.execute("select name, start, end from kernels order by start"))
calls = []; cur = []
for n, s, e in rows:
    if cur and s - cur[-1][2] > 1_000_000: calls.append(cur); cur = []
    cur.append((n, s, e))
if cur: calls.append(cur)
calls = [c for c in calls if len(c) == len(calls[-1])][-30:]
agg = collections.OrderedDict()
for c in calls:
    for i, (n, s, e) in enumerate(c):
        k = (i, n.split("(")[0][-40:])
        a = agg.setdefault(k, [0.0, 0.0, 0])
        a[0] += (e - s) / 1e3; a[1] += ((s - c[i-1][2]) / 1e3 if i else 0.0); a[2] += 1
tot = 0
for (i, n), (d, g, cnt) in agg.items():
    print("%2d %-42s dur %7.1f us   gap before %6.1f us" % (i, n, d/cnt, g/cnt)); tot += (d + g)/cnt
print("first kernel start -> last kernel end: %.1f us over %d calls" % (tot, len(calls)))
PY
