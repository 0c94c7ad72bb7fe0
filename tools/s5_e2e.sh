#!/bin/bash
# end-to-end rate by number of compute streams of the pipeline: tools/s5_e2e.sh <out>
out=gpurun_out/$1; mkdir -p $out
for w in ns p30 c2; do for n in 1 2 3 1 2; do
  HIPSTR_STREAM_COMPUTE_STREAMS=$n timeout 600 python bench.py --workload $w --e2e-only --steps 5 2> $out/${w}_$n.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('end_to_end',d); print('$w streams $n', round(e['alignments_per_s']/1e6,2), 'M/s', round(e['ms_per_pass'],2), 'ms per pass')"
done; done
