#!/bin/bash
# r06_fuzz_fresh.sh — every fuzzer of the repository once more with seeds none of the earlier sweeps used (base seed = $1), on the final
# library: forward path (plain / edges / big, 12 processes), traceback (plain / edges), posteriors + calls, stutter EM, NW + seeds + misc,
# heterogeneous batches.  usage: tools/r06_fuzz_fresh.sh <base seed> [out]
B=${1:-660000}; O=${2:-gpurun_out/r06_fuzz_fresh.txt}; mkdir -p $(dirname $O); : > $O
echo "base seed $B" >> $O
wave(){   # label, count, command prefix (seed appended), suffix
  local label=$1 n=$2 pre=$3 suf=$4; local pids=()
  for i in $(seq 1 $n); do timeout 1500 $pre $((B + 100*i + ${5:-0})) $suf > /tmp/ff_${label}_$i.txt 2>&1 & pids+=($!); done
  for p in "${pids[@]}"; do wait $p; done
  for i in $(seq 1 $n); do echo "$label $i: $(tail -n 1 /tmp/ff_${label}_$i.txt)" >> $O; grep -h "MISMATCH\|refused\|Error\|Traceback" /tmp/ff_${label}_$i.txt | head -5 >> $O; done
}
wave forward 12 "python tools/fuzz_align.py 60" "" 1
wave forward_edges 8 "python tools/fuzz_align.py 25" "edges" 2
wave forward_big 4 "python tools/fuzz_align.py 30" "big" 3
wave forward_tiny 8 "python tools/fuzz_align.py 80" "tiny" 10
wave trace 6 "python tools/fuzz_trace.py 30" "" 4
wave trace_edges 4 "python tools/fuzz_trace.py 10" "edges" 5
wave post 3 "python tools/fuzz_post.py 120" "" 6
wave em 2 "python tools/fuzz_em.py 16" "" 7
wave misc 2 "python tools/fuzz_misc.py 40" "" 8
wave mixed 2 "python tools/fuzz_mixed.py 25" "" 9
cat $O
