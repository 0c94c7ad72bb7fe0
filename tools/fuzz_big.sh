#!/bin/bash
# fuzz_big.sh — the randomised GPU-vs-oracle parity sweep at scale: N processes x M configurations each (the oracle is one CPU thread per
# process), plain and with the re-do path forced on every 3rd / 7th wavefront-allele (HIPSTR_DEBUG_REDO).  usage: tools/fuzz_big.sh [procs] [configs] [out]
NP=${1:-12}; NC=${2:-90}; O=${3:-gpurun_out/fuzz_big.txt}
mkdir -p $(dirname $O); : > $O
run(){   # $1 = label, $2.. = env
  local label=$1; shift
  local pids=()
  for i in $(seq 1 $NP); do
    env "$@" timeout ${FUZZ_TIMEOUT:-1500} python tools/fuzz_align.py $NC $((1000*$i + 17)) > /tmp/fuzz_${label}_$i.txt 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  for i in $(seq 1 $NP); do echo "$label $i $(tail -n 1 /tmp/fuzz_${label}_$i.txt)" >> $O; grep -h "MISMATCH\|refused" /tmp/fuzz_${label}_$i.txt >> $O; done
}
run plain X=1
run redo3 HIPSTR_DEBUG_REDO=3
run redo7 HIPSTR_DEBUG_REDO=7
python - "$O" <<'PY'
import re, sys
tot = bad = 0
for l in open(sys.argv[1]):
    m = re.search(r"alignments (\d+) mismatching configs (\d+)", l)
    if m: tot += int(m.group(1)); bad += int(m.group(2))
print("TOTAL alignments", tot, "mismatching configs", bad)
PY
