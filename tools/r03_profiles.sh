#!/bin/bash
# r03_profiles.sh — every profile summary and bench line of round 3, on the GPU box; results under gpurun_out/r03/, to be copied into profiles/
R=$(pwd); O=$R/gpurun_out/r03; mkdir -p $O
tools/profile_pass.sh r03_ns > $O/pass_ns.log 2>&1
tools/profile_pass.sh r03_c5 --workload c5 > $O/pass_c5.log 2>&1
tools/profile_pass.sh r03_p30 --workload p30 > $O/pass_p30.log 2>&1
tools/profile_pass.sh r03_c4 --workload c4 > $O/pass_c4.log 2>&1
for w in ns c5 p30 c4; do
  for f in kernel_stats.txt sq_counters.json pmc_traffic.json; do cp gpurun_out/prof_r03_$w/$f $O/r03_${w}_$f 2>/dev/null; done
  rm -rf gpurun_out/prof_r03_$w/trace gpurun_out/prof_r03_$w/sq1 gpurun_out/prof_r03_$w/sq2 gpurun_out/prof_r03_$w/fetch gpurun_out/prof_r03_$w/write
done
cd $R
# the counters above are what bench.py reads back: copy them where it looks before the bench lines are taken
for w in ns c5 p30 c4; do for f in sq_counters.json pmc_traffic.json; do cp $O/r03_${w}_$f profiles/ 2>/dev/null; done; done
python bench.py > $O/r03_bench_ns.json 2> $O/bench_ns.err
HIPSTR_SYNTH_IMPERFECT=1.0 python bench.py --no-cpu-baseline > $O/r03_bench_ns_imperfect1.0.json 2> $O/bench_imp.err
for w in c5 p30 c4; do python bench.py --workload $w --no-cpu-baseline > $O/r03_bench_$w.json 2> $O/bench_$w.err; done
ls -la $O
