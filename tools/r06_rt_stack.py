"""Which thread of a streaming process burns CPU beside the library's own (cpu_seconds_by_thread: "python"), and where: p30 passes through the stream
for ~25 s while the parent shell samples per-thread CPU and attaches rocgdb.  usage: python tools/r06_rt_stack.py  (prints its pid first)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from hipstr_amd import capi
print(os.getpid(), flush=True)
hmm = capi.load_hmm(); assert hmm.hipstr_hmm_init(0) == 0
loci, P, A, L, F, sbp, _ = bench.WORKLOADS["p30"]
sb = capi.SynthBatch(n_loci=loci, reads_per_locus=P, n_str_alleles=A, read_len=L, flank_len=F, str_bp=sbp, seed=bench.SEED)
st = capi.Stream(hmm, device=0, slots=8, batch_alignments=0)
probs = np.zeros(sb.n_out); seeds = np.zeros(sb.n_reads, np.int32)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 25
# in-process sampler (ptrace is not permitted on the boxes): a SIGUSR2 handler takes backtrace() on the thread the signal goes to
import ctypes as C, subprocess, collections
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", "/tmp/libts.so", os.path.join(here, "thread_sampler.c"), "-ldl"])
ts = C.CDLL("/tmp/libts.so"); assert ts.ts_init() == 0
def sampler():
    time.sleep(6)
    t0 = bench.thread_cpu_seconds(); time.sleep(3); t1 = bench.thread_cpu_seconds()
    me = threading.get_native_id()
    hot = max((t for t in t1 if t1[t][0] == "python" and t not in (os.getpid(), me)), key=lambda t: t1[t][1] - t0.get(t, ("", 0))[1])
    print("hot thread", hot, "cpu in 3 s:", round(t1[hot][1] - t0[hot][1], 2), flush=True)
    buf = C.create_string_buffer(1 << 14); seen = collections.Counter()
    for i in range(60):
        n = ts.ts_sample(hot, buf, len(buf))
        if n <= 0: seen["(no sample: %d)" % n] += 1
        else: seen[buf.value.decode()] += 1
        time.sleep(0.05)
    for k, v in seen.most_common(6): print("== %d samples\n%s" % (v, k), flush=True)
threading.Thread(target=sampler, daemon=True).start()
n = 0
while time.time() < t_end:
    st.submit_each(sb.ptr); st.flush(); st.collect(loci, probs, seeds); n += 1
print("passes", n, flush=True)
tc = bench.thread_cpu_seconds()
print(sorted(((v[0], t, round(v[1], 2)) for t, v in tc.items()), key=lambda x: -x[2])[:8], flush=True)
st.close()
