#!/bin/bash
# End-to-end (stream) rates per workload with the full host allowance and with one rank's share at 8 GPUs (--host-threads usable/8): gpurun_out/$1/
out=gpurun_out/${1:-r04_e2e}; mkdir -p $out
for wl in p30 c2 ns; do
  for th in 2 0; do
    timeout 900 python bench.py --workload $wl --e2e-only --steps 5 $( [ $th -gt 0 ] && echo --host-threads $th ) > $out/${wl}_t$th.json 2> $out/${wl}_t$th.err
  done
done
