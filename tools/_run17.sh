#!/bin/bash
# kernel trace of the NS workload at imperfect = 1.0 and at the default share
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for imp in 1.0 default; do
  if [ $imp = default ]; then unset HIPSTR_SYNTH_IMPERFECT; else export HIPSTR_SYNTH_IMPERFECT=$imp; fi
  OUT=$R/gpurun_out/kt_$imp; rm -rf $OUT; mkdir -p $OUT
  python $R/bench.py --loci 400 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline > $OUT/bench.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o v -- python $R/bench.py --loci 400 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline > $OUT/trace.log 2>&1
  python $R/tools/rocprof_summary.py $(find $OUT/trace -name '*results.db' | head -1) $OUT/bench.log > $OUT/kernel_stats.txt
  rm -rf $OUT/trace
  head -30 $OUT/kernel_stats.txt
done
