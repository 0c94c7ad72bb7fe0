#!/bin/bash
out=gpurun_out/${1:-r04_slots2}; mkdir -p $out
for wl in ns c2 p30; do
  for cfg in "3 4194304" "6 1048576" "8 2097152" "8 524288"; do
    set -- $cfg
    HIPSTR_BENCH_SLOTS=$1 HIPSTR_BENCH_BATCH=$2 timeout 900 python bench.py --workload $wl --e2e-only --steps 5 > $out/${wl}_t0_s$1_b$2.json 2> $out/${wl}_t0_s$1_b$2.err
  done
done
for cfg in "6 1048576" "8 2097152"; do
  set -- $cfg
  HIPSTR_BENCH_SLOTS=$1 HIPSTR_BENCH_BATCH=$2 timeout 900 python bench.py --workload ns --e2e-only --steps 5 --host-threads 2 > $out/ns_pin2_s$1_b$2.json 2> $out/ns_pin2_s$1_b$2.err
done
