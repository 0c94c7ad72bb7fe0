#!/bin/bash
# the HSA runtime's event thread (a third of a 2-CPU allowance while a stream runs): which setting quiets it?  p30, process pinned to 2 CPUs
for e in "X=1" "ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "HIPSTR_BENCH_BATCH=4194304" "X=2" "ROC_CPU_WAIT_FOR_SIGNAL=0"; do
  echo "== $e"
  env $e timeout 200 python bench.py --workload p30 --steps 5 --e2e-only --host-threads 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['alignments_per_s']/1e6,1), 'M/s', round(d['fraction_of_resident_rate_same_process'],3), 'of resident; cpu us/locus', round(d['process_cpu_us_per_locus'],2), 'batches', d['batches'], d['cpu_seconds_by_thread'][:5])"
done
