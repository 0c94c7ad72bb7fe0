#!/bin/bash
# band shapes of the trailing-flank sweep (experiment build build/variants/libhipstr_hmm_cv.so: HIPSTR_COOP_VARIANT 0 = 4 x 15 (product), 1 = 3 x 20, 2 = 2 x 20, 3 = 3 x 12)
export HIPSTR_HMM_LIB=$PWD/build/variants/libhipstr_hmm_cv.so
for wl in ns p30; do for v in 0 1 2 3 0; do
  echo "== $wl variant $v"
  HIPSTR_COOP_VARIANT=$v python bench.py --workload $wl --steps 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), d['roofline']['phase_ms'])"
done; done
