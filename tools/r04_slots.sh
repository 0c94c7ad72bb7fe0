#!/bin/bash
out=gpurun_out/${1:-r04_slots}; mkdir -p $out
python -m pytest tests/test_posteriors_gpu.py tests/test_expand_gpu.py -x -q -m gpu 2>&1 | tail -3 > $out/pytest.txt
for wl in p30 c2; do
  for cfg in "3 4194304" "6 4194304" "6 1048576" "8 2097152"; do
    set -- $cfg
    HIPSTR_BENCH_SLOTS=$1 HIPSTR_BENCH_BATCH=$2 timeout 900 python bench.py --workload $wl --e2e-only --steps 5 --host-threads 2 > $out/${wl}_pin2_s$1_b$2.json 2> $out/${wl}_pin2_s$1_b$2.err
  done
done
