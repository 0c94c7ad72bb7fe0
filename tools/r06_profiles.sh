#!/bin/bash
# r06_profiles.sh — every profile summary and bench line of round 6, on the GPU box; results under gpurun_out/r06/, to be copied into profiles/.
# Profile passes (tools/profile_pass.sh: kernel trace and every --pmc group in SEPARATE runs): ns, c5, p30, c4, c2, c3 (2000 loci) and — the interrupted-repeat modes of the NS shape (400 loci: HIPSTR_SYNTH_IMPERFECT=1.0, HIPSTR_SYNTH_INHERIT=2).
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O
pass(){   # $1 = tag, rest = bench arguments (environment of the caller applies)
  local tag=$1; shift
  tools/profile_pass.sh r06_$tag "$@" > $O/pass_$tag.log 2>&1
  for f in kernel_stats.txt sq_counters.json pmc_traffic.json; do cp gpurun_out/prof_r06_$tag/$f $O/r06_${tag}_$f 2>/dev/null; done
  rm -rf gpurun_out/prof_r06_$tag/trace gpurun_out/prof_r06_$tag/sq1 gpurun_out/prof_r06_$tag/sq2 gpurun_out/prof_r06_$tag/fetch gpurun_out/prof_r06_$tag/write
}
pass ns
for w in c5 p30 c4 c2; do pass $w --workload $w; done
pass c3 --workload c3 --loci 2000
HIPSTR_SYNTH_IMPERFECT=1.0 pass imperfect --loci 400
HIPSTR_SYNTH_INHERIT=2 pass inherit2 --loci 400
cd $R
# the counters above are what bench.py reads back: copy them where it looks before the bench lines are taken
for w in ns c5 p30 c4 c2 c3 imperfect inherit2; do for f in sq_counters.json pmc_traffic.json; do cp $O/r06_${w}_$f profiles/ 2>/dev/null; done; done
python bench.py > $O/r06_bench_ns.json 2> $O/bench_ns.err
for w in c5 p30 c4 c2; do python bench.py --workload $w --no-cpu-baseline > $O/r06_bench_$w.json 2> $O/bench_$w.err; done
python bench.py --workload c3 --no-cpu-baseline --no-pipeline --steps 3 > $O/r06_bench_c3.json 2> $O/bench_c3.err
HIPSTR_SYNTH_IMPERFECT=1.0 python bench.py --no-cpu-baseline --no-pipeline > $O/r06_bench_ns_imperfect1.0.json 2> $O/bench_imp.err
for k in 1 2 3; do HIPSTR_SYNTH_INHERIT=$k python bench.py --no-cpu-baseline --no-pipeline > $O/r06_bench_ns_inherit$k.json 2> $O/bench_inh$k.err; done
for w in p30 c2 ns; do
  python bench.py --workload $w --e2e-only --steps 5 --host-threads 2 > $O/r06_e2e_${w}_pin2.json 2> $O/e2e_${w}_pin2.err
  python bench.py --workload $w --e2e-only --steps 5 > $O/r06_e2e_${w}_all.json 2> $O/e2e_${w}_all.err
done
python tools/r05_lat.py > $O/r06_latency.txt 2>&1
HIPSTR_FLANK_SYSTOLIC=0 python tools/r05_lat.py >> $O/r06_latency.txt 2>&1
# the one-shot call as bench.py times it (prepared arrays, median of 200) with the host-side buckets; the second per-thread stream off for comparison
LAT_BUCKETS=1 python tools/r05_lat2.py > $O/r06_latency_c.txt 2>&1
python tools/r05_lat2.py >> $O/r06_latency_c.txt 2>&1
bash tools/lat_trace.sh align > $O/r06_lat_trace_align.txt 2>&1
bash tools/lat_trace.sh trace > $O/r06_lat_trace_trace.txt 2>&1
ls -la $O
