#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
for rep in 1 2 3; do
for v in on off; do
  if [ $v = off ]; then export HIPSTR_STR_GROUP_PW=0; else unset HIPSTR_STR_GROUP_PW; fi
  for w in p30 ns; do
    python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > $O/${w}_$v.json 2> $O/${w}_$v.err
    python -c "
import json; d=json.loads([l for l in open('$O/${w}_$v.json') if l.startswith('{')][-1]); e=d['end_to_end']; print('$w $v resident', round(d['value']/1e6,1), 'e2e', round(e['alignments_per_s']/1e6,1), 'frac', round(e['fraction_of_resident_rate'],3), 'worker_host_s', round(e['worker_host_seconds'],2), 'passes', e['passes'], e['one_locus_process_reads_latency']['40x32'])"
  done
done
done
