#!/bin/bash
# kernel trace of the c3 workload (EM + forward + posteriors + calls), per-kernel totals: which kernel the EM's time goes to
R=$(pwd); OUT=$R/gpurun_out/prof_r05_c3; mkdir -p $OUT
ARGS="--workload c3 --loci ${LOCI:-2000} --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline"
cd /tmp && export TMPDIR=/tmp
python $R/bench.py $ARGS > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o v -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) $OUT/bench.log > $OUT/kernel_stats.txt
head -40 $OUT/kernel_stats.txt
