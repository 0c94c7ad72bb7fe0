#!/bin/bash
# end-to-end rate with the adaptive batch size against fixed 2 Mi batches (HIPSTR_STREAM_BIG_BATCH=1): tools/s5_e2e3.sh <out>
out=gpurun_out/$1; mkdir -p $out
for w in ns p30 c2 c5; do for big in 4 1 4 1; do for ht in "" "--host-threads 2"; do
  HIPSTR_STREAM_BIG_BATCH=$big timeout 600 python bench.py --workload $w --e2e-only --steps 5 $ht 2> $out/${w}_$big.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('end_to_end',d); print('$w big x$big $ht', round(e['alignments_per_s']/1e6,2), 'M/s', round(e['ms_per_pass'],2), 'ms per pass', e['batches'], 'batches')"
done; done; done
