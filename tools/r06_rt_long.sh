for e in "HIPSTR_BENCH_E2E_PASSES=150" "HIPSTR_BENCH_E2E_PASSES=150 HIPSTR_BENCH_BATCH=8388608" "HIPSTR_BENCH_E2E_PASSES=150 HIPSTR_BENCH_BATCH=6291456" "HIPSTR_BENCH_E2E_PASSES=150" "HIPSTR_BENCH_E2E_PASSES=150 HIPSTR_BENCH_BATCH=8388608"; do
  echo "== $e"
  env $e timeout 200 python bench.py --workload p30 --steps 5 --e2e-only --host-threads 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['alignments_per_s']/1e6,1), 'M/s', round(d['fraction_of_resident_rate_same_process'],3), 'of resident; cpu us/locus', round(d['process_cpu_us_per_locus'],2), 'batches', d['batches'], 'seconds', round(d['seconds'],2), d['cpu_seconds_by_thread'][:5])"
done
