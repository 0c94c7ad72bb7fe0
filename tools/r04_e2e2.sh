#!/bin/bash
# p30 / c2 end to end: HIPSTR_HOST_THREADS=2 as an environment variable only (the library's pool; feeder, workers, collector unpinned),
# and the whole process pinned to 2 / 4 CPUs
out=gpurun_out/${1:-r04_e2e2}; mkdir -p $out
uptime > $out/uptime.txt
for wl in p30 c2; do
  HIPSTR_HOST_THREADS=2 timeout 900 python bench.py --workload $wl --e2e-only --steps 5 > $out/${wl}_env2.json 2> $out/${wl}_env2.err
  for th in 2 4; do
    timeout 900 python bench.py --workload $wl --e2e-only --steps 5 --host-threads $th > $out/${wl}_pin$th.json 2> $out/${wl}_pin$th.err
  done
done
