#!/bin/bash
# r05_collect.sh — the summaries of a tools/r05_final.sh run (gpurun_out/) into profiles/ (what gets committed)
G=gpurun_out; P=profiles
cp $G/r05/r05_* $P/ 2>/dev/null
cp $G/r05/pytest_gpu.txt $P/r05_pytest_gpu.txt 2>/dev/null
cp $G/flow/flow_rates.txt $P/r05_flow_rates.txt; cp $G/flow/flow_profile.txt $P/r05_flow_profile.txt
cp $G/fuzz_big.txt $P/r05_fuzz.txt; cp $G/fuzz_boundaries.txt $P/r05_fuzz_boundaries.txt; cp $G/r05_uptime.txt $P/r05_uptime.txt
ls $P | grep -c r05_
