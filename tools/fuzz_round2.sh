#!/bin/bash
# fuzz_round2.sh — the round's new sweeps once more with seeds of their own (mixed batches, posteriors + genotype calls, stutter EM, NW + seeds)
O=${1:-gpurun_out/fuzz_round2.txt}; mkdir -p $(dirname $O); : > $O
pids=()
for i in 1 2 3 4 5 6; do timeout 1200 python tools/fuzz_mixed.py 40 $((9500 + i)) > /tmp/r2m_$i.txt 2>&1 & pids+=($!); done
for i in 1 2 3 4; do timeout 1200 python tools/fuzz_post.py 150 $((20 + i)) > /tmp/r2p_$i.txt 2>&1 & pids+=($!); done
for i in 1 2 3; do timeout 1200 python tools/fuzz_em.py 16 $((20 + i)) > /tmp/r2e_$i.txt 2>&1 & pids+=($!); done
for i in 1 2 3; do timeout 1200 python tools/fuzz_misc.py 80 $((20 + i)) > /tmp/r2x_$i.txt 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
for i in 1 2 3 4 5 6; do echo "mixed $i: $(tail -n 1 /tmp/r2m_$i.txt)" >> $O; grep -h MISMATCH /tmp/r2m_$i.txt | head -3 >> $O; done
for i in 1 2 3 4; do echo "post $i: $(tail -n 1 /tmp/r2p_$i.txt)" >> $O; grep -h MISMATCH /tmp/r2p_$i.txt | head -3 >> $O; done
for i in 1 2 3; do echo "em $i: $(tail -n 1 /tmp/r2e_$i.txt)" >> $O; grep -h MISMATCH /tmp/r2e_$i.txt | head -3 >> $O; done
for i in 1 2 3; do echo "nw+seeded $i: $(tail -n 1 /tmp/r2x_$i.txt)" >> $O; grep -h MISMATCH /tmp/r2x_$i.txt | head -3 >> $O; done
cat $O
