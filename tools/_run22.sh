#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
HIPSTR_TIMING=1 python bench.py --workload p30 --no-cpu-baseline --steps 2 --warmup 1 > $O/p30_t.json 2> $O/p30_t.err
grep -c . $O/p30_t.err
grep -E "hipstr_hmm_upload: total|stream:" $O/p30_t.err | tail -30
grep -E "^prepare_batch" $O/p30_t.err | tail -12
