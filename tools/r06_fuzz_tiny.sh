#!/bin/bash
# r06_fuzz_tiny.sh — the tiny-allele shapes (two to a dozen copies of period-1..3 motifs) through the forward path and the traceback.  usage: tools/r06_fuzz_tiny.sh <base seed> [out]
B=${1:-880000}; O=${2:-gpurun_out/r06_fuzz_tiny.txt}; mkdir -p $(dirname $O); : > $O
echo "base seed $B" >> $O
pids=()
for i in 1 2 3 4 5 6 7 8; do timeout 1200 python tools/fuzz_align.py 100 $((B + 100*i)) tiny > /tmp/ft_a_$i.txt 2>&1 & pids+=($!); done
for i in 1 2 3 4 5 6; do timeout 1200 python tools/fuzz_trace.py 50 $((B + 100*i + 7)) tiny > /tmp/ft_t_$i.txt 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
for i in 1 2 3 4 5 6 7 8; do echo "forward tiny $i: $(tail -n 1 /tmp/ft_a_$i.txt)" >> $O; grep -h "MISMATCH\|refused\|Error" /tmp/ft_a_$i.txt | head -5 >> $O; done
for i in 1 2 3 4 5 6; do echo "traceback tiny $i: $(tail -n 1 /tmp/ft_t_$i.txt)" >> $O; grep -h "MISMATCH\|refused\|Error" /tmp/ft_t_$i.txt | head -5 >> $O; done
cat $O
