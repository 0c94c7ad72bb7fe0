#!/bin/bash
# end-to-end rate by batch size / slots of the pipeline: tools/r04_e2e_batch.sh <out>
out=gpurun_out/$1; mkdir -p $out
for w in ns; do for cfg in "2 8" "4 8" "4 4" "6 4" "8 3" "2 8" "4 6"; do
  set -- $cfg
  HIPSTR_BENCH_BATCH=$(( $1 << 20 )) HIPSTR_BENCH_SLOTS=$2 timeout 600 python bench.py --workload $w --e2e-only --steps 5 2> $out/${w}_$1_$2.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('end_to_end',d); print('$w batch $1 Mi slots $2', round(e['alignments_per_s']/1e6,2), 'M/s', round(e['ms_per_pass'],2), 'ms per pass', e['batches'], 'batches')"
done; done
