#!/bin/bash
# fuzz_more.sh — a second, independently seeded sweep: forward path (tools/fuzz_align.py) and traceback (tools/fuzz_trace.py: incl. inherited
# interruptions and read sides of up to 1024 columns), N processes each, every process under a timeout.  usage: tools/fuzz_more.sh [procs] [configs] [out]
NP=${1:-12}; NC=${2:-60}; O=${3:-gpurun_out/fuzz_more.txt}
mkdir -p $(dirname $O); : > $O
pids=()
for i in $(seq 1 $NP); do
  ( timeout 1200 python tools/fuzz_align.py $NC $((7000*$i + 31)) > /tmp/fa_$i.txt 2>&1; echo "align $i $(tail -n 1 /tmp/fa_$i.txt)" >> $O; grep -h "MISMATCH" /tmp/fa_$i.txt >> $O ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
pids=()
for i in $(seq 1 $NP); do
  ( timeout 1200 python tools/fuzz_trace.py $NC $((9000*$i + 13)) > /tmp/ft_$i.txt 2>&1; echo "trace $i $(tail -n 1 /tmp/ft_$i.txt)" >> $O; grep -h "MISMATCH" /tmp/ft_$i.txt >> $O ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
python - "$O" <<'PY'
import re, sys
a = t = bad = 0
for l in open(sys.argv[1]):
    m = re.search(r"alignments (\d+) mismatching configs (\d+)", l)
    if m: a += int(m.group(1)); bad += int(m.group(2))
    m = re.search(r"tracebacks (\d+) mismatching configs (\d+)", l)
    if m: t += int(m.group(1)); bad += int(m.group(2))
print("TOTAL alignments", a, "tracebacks", t, "mismatching configs", bad)
PY
