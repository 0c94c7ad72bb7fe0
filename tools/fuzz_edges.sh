#!/bin/bash
# fuzz_edges.sh — the boundary sweeps of round 5 in one call: forward path and traceback with reads / alleles per locus around the packing sizes (8 + 4 processes),
# posterior + genotype calls and the stutter EM at their size boundaries.  usage: tools/fuzz_edges.sh [out]
O=${1:-gpurun_out/fuzz_boundaries.txt}; mkdir -p $(dirname $O); : > $O
pids=()
for i in 1 2 3 4 5 6 7 8; do timeout 1200 python tools/fuzz_align.py ${FUZZ_N:-25} $((7000 + i)) edges > /tmp/fe_$i.txt 2>&1 & pids+=($!); done      # (waited for by PID)
for p in "${pids[@]}"; do wait $p; done
for i in 1 2 3 4 5 6 7 8; do echo "forward edges $i: $(tail -n 1 /tmp/fe_$i.txt)" >> $O; grep -h "MISMATCH\|refused\|Error" /tmp/fe_$i.txt | head -5 >> $O; done
pids=()
for i in 1 2 3 4; do timeout 1200 python tools/fuzz_trace.py ${FUZZ_NT:-10} $((8000 + i)) edges > /tmp/ft_$i.txt 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
for i in 1 2 3 4; do echo "traceback edges $i: $(tail -n 1 /tmp/ft_$i.txt)" >> $O; grep -h "MISMATCH\|refused\|Error" /tmp/ft_$i.txt | head -5 >> $O; done
for s in 11 12 13; do echo "posteriors + genotype calls, seed $s: $(timeout 900 python tools/fuzz_post.py 120 $s 2>&1 | tail -n 3 | tr '\n' ' ')" >> $O; done
for s in 11 12; do echo "stutter EM, seed $s: $(timeout 900 python tools/fuzz_em.py 16 $s 2>&1 | tail -n 3 | tr '\n' ' ')" >> $O; done
cat $O
