#!/bin/bash
out=gpurun_out/${1:-r04_wt}; mkdir -p $out
for wl in p30 c2 ns; do for wt in 1 2 5; do
  HIPSTR_STREAM_WORKER_THREADS=$wt timeout 900 python bench.py --workload $wl --e2e-only --steps 5 > $out/${wl}_wt$wt.json 2> $out/${wl}_wt$wt.err
done; done
