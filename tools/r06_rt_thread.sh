#!/bin/bash
# the HIP-runtime thread that is busy for a third of a 2-CPU allowance (cpu_seconds_by_thread: "python"): does a runtime setting quiet it?
for e in "X=1" "ROC_ACTIVE_WAIT_TIMEOUT=0" "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=0 GPU_MAX_HW_QUEUES=2" "X=2"; do
  echo "== $e"
  env $e timeout 200 python bench.py --workload p30 --steps 5 --e2e-only --host-threads 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['alignments_per_s']/1e6,1), 'M/s', round(d['fraction_of_resident_rate_same_process'],3), 'of resident; cpu us/locus', round(d['process_cpu_us_per_locus'],2), d['cpu_seconds_by_thread'][:5])"
done
