#!/bin/bash
# p30_e2e.sh — the production-like shape end to end through the stream, a few times (the boxes are noisy): resident rate, end-to-end rate, fraction
for rep in 1 2 3 4; do
python bench.py --workload p30 --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d['end_to_end']; print('p30 resident', round(d['value']/1e6,1), 'e2e', round(e['alignments_per_s']/1e6,1), 'frac', round(e['fraction_of_resident_rate'],3), 'worker_host_s', round(e['worker_host_seconds'],3), 'batches', e['batches'], 'lat', round(e['one_locus_process_reads_latency']['40x32']['median_ms'],3))"
done
