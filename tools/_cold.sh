#!/bin/bash
cat /proc/loadavg
for w in 8 1 1; do
HIPSTR_BENCH_E2E_WARMUP=$w python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); e=d.get('end_to_end') or {}; print('warmup $w resident', round(d['value']/1e6,1), 'e2e', round(e.get('alignments_per_s',0)/1e6,1), e.get('ms_per_pass'))"
done
