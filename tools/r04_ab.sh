#!/bin/bash
# A/B on one box: blocking vs spinning collectors, alternating
out=gpurun_out/${1:-r04_ab}; mkdir -p $out
uptime > $out/uptime.txt
for rep in 1 2; do
for wl in p30 ns; do
  for th in 2 0; do
    for spin in 0 1; do
      HIPSTR_STREAM_SPIN=$spin timeout 900 python bench.py --workload $wl --e2e-only --steps 5 $( [ $th -gt 0 ] && echo --host-threads $th ) > $out/${wl}_t${th}_spin${spin}_r$rep.json 2> $out/${wl}_t${th}_spin${spin}_r$rep.err
    done
  done
done
done
uptime >> $out/uptime.txt
