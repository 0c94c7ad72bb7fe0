#!/bin/bash
# fuzz_around.sh — round 5's sweeps beside the forward fuzz: heterogeneous batches (tools/fuzz_mixed.py: loci of different shapes in one call and
# through the stream) and the stages around the path (tools/fuzz_misc.py: Needleman-Wunsch at tile boundaries, caller-chosen seeds in the
# forward path and the traceback).  usage: tools/fuzz_around.sh [out]
O=${1:-gpurun_out/fuzz_around.txt}; mkdir -p $(dirname $O); : > $O
pids=()
for i in 1 2 3 4 5 6; do timeout 1500 python tools/fuzz_mixed.py ${FUZZ_NM:-30} $((9000 + i)) > /tmp/fm_$i.txt 2>&1 & pids+=($!); done
for i in 1 2 3 4; do timeout 1500 python tools/fuzz_misc.py ${FUZZ_NX:-60} $((9100 + i)) > /tmp/fx_$i.txt 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
for i in 1 2 3 4 5 6; do echo "mixed batches $i: $(tail -n 1 /tmp/fm_$i.txt)" >> $O; grep -h "MISMATCH\|refused\|Error" /tmp/fm_$i.txt | head -5 >> $O; done
for i in 1 2 3 4; do echo "nw + seeded $i: $(tail -n 1 /tmp/fx_$i.txt)" >> $O; grep -h "MISMATCH\|Error" /tmp/fx_$i.txt | head -5 >> $O; done
cat $O
