#!/bin/bash
# r04_profiles.sh — every profile summary and bench line of round 4, on the GPU box; results under gpurun_out/r04/, to be copied into profiles/
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O
for w in ns c5 p30 c4 c2; do
  if [ $w = ns ]; then tools/profile_pass.sh r04_$w > $O/pass_$w.log 2>&1; else tools/profile_pass.sh r04_$w --workload $w > $O/pass_$w.log 2>&1; fi
  for f in kernel_stats.txt sq_counters.json pmc_traffic.json; do cp gpurun_out/prof_r04_$w/$f $O/r04_${w}_$f 2>/dev/null; done
  rm -rf gpurun_out/prof_r04_$w/trace gpurun_out/prof_r04_$w/sq1 gpurun_out/prof_r04_$w/sq2 gpurun_out/prof_r04_$w/fetch gpurun_out/prof_r04_$w/write
done
cd $R
# the counters above are what bench.py reads back: copy them where it looks before the bench lines are taken
for w in ns c5 p30 c4 c2; do for f in sq_counters.json pmc_traffic.json; do cp $O/r04_${w}_$f profiles/ 2>/dev/null; done; done
python bench.py > $O/r04_bench_ns.json 2> $O/bench_ns.err
for w in c5 p30 c4 c2; do python bench.py --workload $w --no-cpu-baseline > $O/r04_bench_$w.json 2> $O/bench_$w.err; done
python bench.py --workload c3 --no-cpu-baseline --no-pipeline --steps 3 > $O/r04_bench_c3.json 2> $O/bench_c3.err
HIPSTR_SYNTH_IMPERFECT=1.0 python bench.py --no-cpu-baseline --no-pipeline > $O/r04_bench_ns_imperfect1.0.json 2> $O/bench_imp.err
for k in 1 2 3; do HIPSTR_SYNTH_INHERIT=$k python bench.py --no-cpu-baseline --no-pipeline > $O/r04_bench_ns_inherit$k.json 2> $O/bench_inh$k.err; done
# the host share: environment variable only (the library's pool; feeder, workers, collector unpinned) and the whole process pinned
for w in p30 c2 ns; do
  HIPSTR_HOST_THREADS=2 python bench.py --workload $w --e2e-only --steps 5 > $O/r04_e2e_${w}_env2.json 2> $O/e2e_${w}_env2.err
  python bench.py --workload $w --e2e-only --steps 5 --host-threads 2 > $O/r04_e2e_${w}_pin2.json 2> $O/e2e_${w}_pin2.err
  python bench.py --workload $w --e2e-only --steps 5 > $O/r04_e2e_${w}_all.json 2> $O/e2e_${w}_all.err
done
ls -la $O
